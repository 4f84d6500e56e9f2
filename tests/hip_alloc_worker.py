"""``AKS_HOST_ALLOC=hip``: the drop-in with NO torch in the process -- device memory, stream, events and pinned staging
through the HIP runtime alone (arnoldi_amd/mem.py).  Solves BASELINE config 1 (mark(50), LR) and the smoke solve's planted
random graph in the binned form against the CPU oracle, plus the explicit-restart solver and device-side residuals, and
reports whether torch ever got imported.

    AKS_HOST_ALLOC=hip python tests/hip_alloc_worker.py OUT.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arnoldi-py_amd")):
    sys.path.insert(0, p)
assert os.environ.get("AKS_HOST_ALLOC") == "hip"

import numpy as np  # noqa: E402


def main(out_path):
    import arnoldi_amd
    import oracle
    from arnoldi_amd import mem
    from arnoldi_amd.engine import CsrOperator
    from arnoldi_amd.explicit_restarts import explicit_restarts_with_deflation
    from arnoldi_amd.matrices import mark, random_csr

    res = {"backend": mem.BACKEND}
    # the array layer itself against numpy: pitched 2-D views, row / block slices, strided copies in both directions,
    # views, zeroing, pinned staging
    dev = mem.as_device(None)
    rng = np.random.default_rng(0)
    host = (rng.standard_normal((7, 640)) + 1j * rng.standard_normal((7, 640))).astype(np.complex128)
    a = mem.zeros((7, 640), mem.c128, dev)
    want = np.zeros_like(host)
    steps = {}
    a[1:6, :600].copy_(mem.host(host[1:6, :600]))                              # host -> pitched block
    want[1:6, :600] = host[1:6, :600]
    steps["block_upload"] = np.array_equal(a.cpu().numpy(), want)
    a[3].zero_()
    want[3] = 0
    a[0, :5].copy_(mem.host(host[0, :5]))
    want[0, :5] = host[0, :5]
    steps["row_zero_and_row_slice"] = np.array_equal(a.cpu().numpy(), want)
    b = mem.empty((4, 100), mem.c128, dev)
    b.copy_(a[2:6, 50:150])                                                    # strided device -> contiguous device
    want_b = want[2:6, 50:150].copy()
    steps["strided_to_contiguous"] = np.array_equal(b.cpu().numpy(), want_b) and np.array_equal(a.cpu().numpy(), want)
    a[0:4, 10:110].copy_(b)                                                    # contiguous -> strided
    want[0:4, 10:110] = want_b
    steps["contiguous_to_strided"] = np.array_equal(a[0:4, 10:110].cpu().numpy(), want_b) and np.array_equal(a.cpu().numpy(), want)
    a[4:6, 600:640].zero_()                                                    # 2-D memset
    a[1:3, 0:8].zero_()
    want[4:6, 600:640] = 0
    want[1:3, 0:8] = 0
    steps["block_zero"] = np.array_equal(a.cpu().numpy(), want)
    flat = mem.upload(np.arange(32, dtype=np.float64), dev)
    steps["flat_views"] = (flat.view(mem.c128).numel() == 16 and flat[5].item() == 5.0
                           and flat[3:9].cpu().numpy().tolist() == [3, 4, 5, 6, 7, 8]
                           and flat.view(mem.c128)[2:4].cpu().numpy().tolist() == [4 + 5j, 6 + 7j])
    pin = mem.pinned_empty((4, 100), mem.c128)
    pin.copy_(b, non_blocking=True)
    ev = mem.Event()
    ev.record()
    ev.synchronize()
    steps["pinned_download"] = np.array_equal(pin.numpy(), want_b)
    pin.numpy()[:] = 1.5
    b.copy_(pin, non_blocking=True)
    mem.synchronize()
    steps["pinned_upload"] = bool((b.cpu().numpy() == 1.5).all())
    res["array_layer"] = {k: bool(v) for k, v in steps.items()}
    ok = all(steps.values())
    res["array_layer_ok"] = bool(ok)
    A = mark(50)
    kw = dict(max_dim=20, stopping_criterion=1e-8, sort_function=oracle.arg_largest_real)
    np.random.seed(0)
    st = {}
    Q, T, hist = arnoldi_amd.partial_schur(A, 5, stats=st, **kw)
    np.random.seed(0)
    Qo, To, histo = oracle.krylov_schur(A, 5, **kw)
    _, _, rel = oracle.eig_residuals(A, Q, T)
    _, _, rel_o = oracle.eig_residuals(A, Qo, To)
    _, _, drel = st["solver"].true_residuals()
    res["mark50"] = {"hist_equal": bool(np.array_equal(hist.restarts, histo.restarts) and np.array_equal(hist.matvecs, histo.matvecs)),
                     "eig_err": float(np.abs(np.diag(T) - np.diag(To)).max()), "rel": float(rel.max()), "rel_oracle": float(rel_o.max()),
                     "device_residual_err": float(np.abs(np.sort(drel) - np.sort(rel)).max()), "restarts": int(st["restarts"])}
    B = random_csr(20000, 5, 1234, planted=(4.0, 3.7, 3.4, 3.1, 2.8, 2.5))
    np.random.seed(1)
    st2 = {}
    Q2, T2, h2 = arnoldi_amd.partial_schur(CsrOperator(B, spmv_form="binned"), 5, max_dim=20, stats=st2)
    np.random.seed(1)
    Qo2, To2, ho2 = oracle.krylov_schur(B, 5, max_dim=20)
    _, _, rel2 = oracle.eig_residuals(B, Q2, T2)
    _, _, rel_o2 = oracle.eig_residuals(B, Qo2, To2)
    res["binned"] = {"hist_equal": bool(np.array_equal(h2.restarts, ho2.restarts)), "form": st2["spmv_form"],
                     "deferred": int(st2["deferred_normalisations"]), "rel": float(rel2.max()), "rel_oracle": float(rel_o2.max())}
    # real-packed mode and the explicit-restart solver on the same backend
    np.random.seed(0)
    Q3, T3, h3 = arnoldi_amd.partial_schur(A, 5, arithmetic="real", **kw)
    _, _, rel3 = oracle.eig_residuals(A, Q3, T3)
    res["real"] = {"rel": float(rel3.max()), "eig_err": float(np.abs(np.sort_complex(np.linalg.eigvals(T3)) - np.sort_complex(np.diag(To).astype(complex))).max())}
    M = mark(30)
    kw30 = dict(max_dim=30, stopping_criterion=1e-8, sort_function=oracle.arg_largest_real)
    np.random.seed(1)
    vals, vecs, h5 = explicit_restarts_with_deflation(M, 4, **kw30)
    np.random.seed(1)
    vo, xo, ho = oracle.explicit_restarts_with_deflation(M, 4, **kw30)
    res["deflation"] = {"hist_equal": bool(np.array_equal(h5.restarts, ho.restarts)), "eig_err": float(np.abs(vals - vo).max())}
    # hipGraph replay through the runtime's own capture API (mem.Graph): the re-expansions of the same solve captured
    # once and replayed -- same bits as the eager sequence, and a graph was really built
    graph = {}
    for flag in ("0", "1"):
        os.environ["AKS_GRAPH"] = flag
        np.random.seed(1)
        stg = {}
        Qg, Tg, hg = arnoldi_amd.partial_schur(B, 5, max_dim=20, stats=stg)
        graph[flag] = (Qg, Tg, hg.restarts.copy(), len(stg["solver"].ctx._graphs), int(stg["restarts"]))
    del os.environ["AKS_GRAPH"]
    res["graph"] = {"bit_identical": bool(np.array_equal(graph["0"][0], graph["1"][0]) and np.array_equal(graph["0"][1], graph["1"][1])
                                          and np.array_equal(graph["0"][2], graph["1"][2])),
                    "graphs_eager": graph["0"][3], "graphs_replayed": graph["1"][3], "restarts": graph["1"][4]}
    res["torch_imported"] = "torch" in sys.modules
    json.dump(res, open(out_path, "w"))
    print(res)


if __name__ == "__main__":
    main(sys.argv[1])
