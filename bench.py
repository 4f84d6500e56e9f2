#!/usr/bin/env python3
"""Benchmark of the hot path: Krylov restarts / second (+ SpMV GB/s vs the HBM roofline).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "k=5 m=20", configs[4] -- fits one GPU): synthetic random CSR,
n = 10,000,000 rows, 5 non-zeros per row (nnz ~= 50M), float64 values, default_rng(1234);
nev = 5, max_dim = 20, p = 10, start vector np.random.seed(0).  A "step" is one steady-state
Krylov-Schur restart: host Schur + reorder of the 20x20 projected matrix, the truncation
V[:, :10] = V[:, :20] Qp, and 10 Arnoldi steps (SpMV + DGKS Gram-Schmidt) back to width 20.
The matrix has no dominant eigenvalues, so the solve never converges: K restarts are timed.
For N > 1 the same n = 10M problem is row-sharded over the ranks (strong scaling).

The line printed by rank 0 also carries
  roofline      live HIP-event timing of the SpMV kernel launches inside the timed region,
                algorithmic bytes (12 nnz + 36 n + 4) / average launch time vs 8 TB/s;
  roofline_ortho  the same for the Gram-Schmidt launches of the timed region;
  cpu_baseline  the CPU oracle (NumPy/SciPy restatement of the reference, validated against
                the reference's golden outputs) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "arnoldi-py_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CONFIG_OF = {"random": 5, "laplace2d": 2, "laplace3d": 4, "markov": 1, "banded": 3}   # BASELINE.json configs[] (1-based)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--rows", dest="n", type=int, default=10_000_000,
                    help="matrix dimension (use --rows under torch.distributed.run, whose own parser "
                         "treats --n as an ambiguous abbreviation)")
    ap.add_argument("--per-row", type=int, default=5)
    ap.add_argument("--nev", type=int, default=5)
    ap.add_argument("--max-dim", type=int, default=20)
    ap.add_argument("--workload", choices=["random", "laplace2d", "laplace3d", "markov", "banded"],
                    default="random",
                    help="random = BASELINE config 5 (default); laplace2d / laplace3d = configs 2 / 4; "
                         "markov = mark(M) of the reference's README scaled to ~n rows (sorted LR); "
                         "banded = stand-in for config 3 (af_shell10 is not available offline): use "
                         "--n 1500000 --per-row 35 --nev 20 --max-dim 41")
    ap.add_argument("--cpu-sample-n", type=int, default=1_000_000)
    ap.add_argument("--cpu-restarts", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-real-leg", action="store_true",
                    help="skip the extra measurement of partial_schur(arithmetic='real') on the same matrix")
    ap.add_argument("--real-leg", action="store_true",
                    help="run that extra measurement with several GPUs too (default: one GPU only, so that "
                         "nothing after the main measurement can cost a multi-GPU run its JSON line)")
    ap.add_argument("--chained", action="store_true",
                    help="force the Python-chained stage path on one GPU (the multi-GPU code path)")
    return ap.parse_args()


def grid_dims(workload, n):
    if workload == "markov":                      # mark(M): n = M (M + 1) / 2
        return (int(round((2 * n) ** 0.5)),)
    if workload == "laplace2d":
        nx = int(round(n ** 0.5))
        return (nx, nx + 1)
    nx = int(round(n ** (1.0 / 3.0)))
    return (nx, nx + 1, nx + 2)


def build_rows(args, r0, r1, n, dims):
    from arnoldi_amd import matrices

    if args.workload == "random":
        return matrices.random_csr(n, args.per_row, 1234, row_range=(r0, r1))
    if args.workload == "banded":
        return matrices.banded_csr(n, args.per_row, 1234)[r0:r1]
    if args.workload == "markov":
        rows = matrices.mark(dims[0])[r0:r1]      # every rank builds the chain and keeps its rows
    else:
        rows = matrices.laplace_rows(dims, r0, r1)
    assert rows.shape == (r1 - r0, n)
    return rows


def ortho_algorithmic_bytes(n_local, J, second):
    """Bytes the fused Gram-Schmidt schedule must move for one step at panel width J:
    panel reads (projection, update+re-projection, second update if run) and the w traffic
    of each stage (fused figure)."""
    panel = 16 * n_local * J * (3 if second else 2)
    w = 16 * n_local * (1 + 2 + (2 if second else 0) + 2)
    return panel + w


def cpu_baseline(args, n_full):
    """Steady-state restarts/s of the CPU oracle on a bounded sample of the same workload."""
    import scipy.sparse as sp  # noqa: F401
    import oracle
    from arnoldi_amd import matrices

    ns = min(args.cpu_sample_n, n_full)
    if args.workload == "random":
        A = matrices.random_csr(ns, args.per_row, 1234)
        what = f"random CSR n={ns} ({args.per_row}/row, same generator)"
    elif args.workload == "banded":
        A = matrices.banded_csr(ns, args.per_row, 1234)
        what = f"banded CSR n={ns} ({args.per_row}/row, same generator)"
    elif args.workload == "markov":
        mm = grid_dims("markov", ns)[0]
        A = matrices.mark(mm)
        ns = A.shape[0]
        what = f"mark({mm})"
    else:
        dims = grid_dims(args.workload, ns)
        ns = int(np.prod(dims))
        A = matrices.laplace_rows(dims, 0, ns)
        what = f"{args.workload} grid {dims}"
    A = A.astype(np.complex128)  # as the reference's scripts do (benchmark-partial-schur.py:78)
    np.random.seed(0)
    trace = {}
    t0 = time.perf_counter()
    try:
        oracle.krylov_schur(A, args.nev, max_dim=args.max_dim, max_restarts=args.cpu_restarts, trace=trace,
                            sort_function=oracle.arg_largest_real if args.workload == "markov" else None)
    except ValueError:
        pass
    wall = time.perf_counter() - t0
    ts = trace.get("t_restart", [])
    if len(ts) < 2:
        return None
    per_restart = (ts[-1] - ts[0]) / (len(ts) - 1)
    scale = ns / n_full
    try:
        from threadpoolctl import threadpool_info

        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = len(os.sched_getaffinity(0))
    # per-phase split of the CPU path at the sample size: one operator apply and one
    # dgks_gs call at the mean panel width of a restart
    rng = np.random.default_rng(0)
    xs = (rng.standard_normal(ns) + 1j * rng.standard_normal(ns)).astype(np.complex128)
    t0 = time.perf_counter()
    for _ in range(3):
        oracle.csr_matvec(A, xs)
    ms_matvec = (time.perf_counter() - t0) / 3 * 1e3
    p_ = min(args.nev + 5, args.max_dim - 1)
    Jm = (p_ + 1 + args.max_dim) // 2
    Vp, _ = np.linalg.qr(rng.standard_normal((ns, Jm)) + 0j)
    Vp = np.asfortranarray(Vp)
    t0 = time.perf_counter()
    for _ in range(3):
        oracle.dgks_gs(xs.copy(), Vp, np.zeros(Jm, np.complex128), 1e-8)
    ms_dgks = (time.perf_counter() - t0) / 3 * 1e3
    return {
        "value": (1.0 / per_restart) * scale,
        "unit": "restarts/s",
        "cores": int(blas_threads),
        "ms_per_matvec_at_sample_n": round(ms_matvec, 2),
        "ms_per_dgks_gs_at_sample_n": round(ms_dgks, 2),
        "dgks_panel_width": Jm,
        "kind": "port",
        "sample": (f"oracle.krylov_schur on {what}, A.astype(complex128), {len(ts) - 1} steady-state "
                   f"restarts timed ({per_restart:.3f} s each, {wall:.1f} s CPU wall in all); all work is "
                   f"O(n), so the rate is scaled by {ns}/{n_full}; SciPy SpMV is single-threaded, "
                   f"BLAS uses {blas_threads} threads of {os.cpu_count()} host CPUs"),
        "measured_restart_s_at_sample_n": per_restart,
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # AKS_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path with all ranks sharing the visible GPU(s)
    # (collectives staged through host memory); the measured numbers then mean nothing.
    backend = os.environ.get("AKS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)

    import torch.distributed as dist
    from arnoldi_amd import _hip
    from arnoldi_amd.dist import Comm, row_offsets
    from arnoldi_amd.engine import CsrOperator
    from arnoldi_amd.krylov_schur import KrylovSchurSolver
    from arnoldi_amd.utils import arg_largest_magnitude, arg_largest_real

    comm = None
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        comm = Comm()
    elif os.environ.get("AKS_FORCE_COMM") == "1":
        # rehearsal of the multi-rank host path on one GPU: a one-rank RCCL group whose
        # all-reduces are really issued (measures the per-step host + collective latency)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        comm = Comm(force=True)

    n, dims = args.n, None
    if args.workload not in ("random", "banded"):
        dims = grid_dims(args.workload, args.n)      # computed ONCE: n below is the grid's row count
        n = dims[0] * (dims[0] + 1) // 2 if args.workload == "markov" else int(np.prod(dims))
    offsets = row_offsets(n, world)
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    t_setup = time.perf_counter()
    rows = build_rows(args, r0, r1, n, dims)
    op = CsrOperator(local_rows=rows, offsets=offsets, comm=comm)
    nnz_local = op.nnz

    nev, m = args.nev, args.max_dim
    p = min(nev + 5, m - 1)
    np.random.seed(0)
    sort_key = arg_largest_real if args.workload == "markov" else arg_largest_magnitude
    solver = KrylovSchurSolver(op, nev, m, p, 1e-8, sort_key, comm=comm)
    ctx = solver.ctx
    native = world == 1 and not args.chained and comm is None
    if not native:
        ctx.force_chained = True
    t_setup = time.perf_counter() - t_setup

    def sync():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()

    sync()
    t0 = time.perf_counter()
    assert solver.start() == m
    torch.cuda.synchronize()
    initial_ms = (time.perf_counter() - t0) * 1e3

    for i in range(args.warmup):
        solver.contract(i)
        solver.expand()

    probe = None
    if native:
        probe = _hip.Probe(capacity=2 * (m - p) * args.steps + 8)
        ctx.probe = probe
    else:
        ctx.spmv_events = []
    second0 = int(ctx.last_ctrl.second_passes)
    steps0 = int(ctx.last_ctrl.steps_done)

    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        solver.contract(args.warmup + i)
        solver.expand()
    sync()
    elapsed = time.perf_counter() - t0

    if comm is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if comm.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel device time measured inside the timed region --------------------------
    if probe is not None:
        n_spmv, spmv_ms = probe.read(_hip.PROBE_SPMV)
        n_ortho, ortho_ms = probe.read(_hip.PROBE_ORTHO)
    else:
        ev = ctx.spmv_events
        n_spmv, spmv_ms = len(ev), sum(a.elapsed_time(b) for a, b in ev)
        n_ortho, ortho_ms = 0, 0.0
    spmv_avg_ms = spmv_ms / max(n_spmv, 1)
    spmv_bytes = op.algorithmic_bytes()
    achieved = spmv_bytes / (spmv_avg_ms * 1e-3) / 1e9 if n_spmv else None

    steps_done = int(ctx.last_ctrl.steps_done) - steps0
    seconds = int(ctx.last_ctrl.second_passes) - second0
    ortho = None
    frac_second = seconds / max(steps_done, 1)
    per_cycle = 0.0   # Gram-Schmidt bytes of one restart's m - p steps, at the measured second-pass rate
    for J in range(p + 1, m + 1):
        per_cycle += (frac_second * ortho_algorithmic_bytes(op.n_local, J, True)
                      + (1 - frac_second) * ortho_algorithmic_bytes(op.n_local, J, False))
    if n_ortho:
        total = per_cycle * args.steps
        a = total / (ortho_ms * 1e-3) / 1e9
        ortho = {"bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(a / HBM_PEAK_GBS, 4), "launch_groups": n_ortho,
                 "avg_ms_per_step": round(ortho_ms / n_ortho, 4),
                 "second_pass_fraction": round(frac_second, 3), "traffic": None}

    if rank == 0:
        # HBM bytes per SpMV from the committed rocprofv3 --pmc passes of this same command on the
        # default workload (profiles/collect_pmc.sh -> profiles/pmc_summary.json): FETCH_SIZE is
        # doubled for the coalesced streams of the binned kernels (gfx950 counts their 128-B
        # requests as 64 B, MI355X_MICROARCH.md "HBM"); the CSR kernel's 16-B gathers are
        # reported raw (64-B requests).
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
        default_workload = args.workload == "random" and n == 10_000_000 and world == 1
        if os.path.exists(pmc_path) and default_workload:
            try:
                pmc = json.load(open(pmc_path))
                if op.spmv_form == "binned":
                    traffic = (pmc["k_pb_phase1"]["hbm_bytes_per_launch_fetch_x2"]
                               + pmc["k_pb_phase2"]["hbm_bytes_per_launch_fetch_x2"])
                else:
                    traffic = pmc["k_spmv"]["hbm_bytes_per_launch_raw"]
            except Exception:
                traffic = None
        out = {
            "metric": "krylov_restarts_per_sec",
            "value": round(args.steps / elapsed, 4),
            "unit": "restarts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "complex128",
            "data": "synthetic",
            "config": {
                "workload": (f"{args.workload} CSR n={n} nnz={nnz_local if world == 1 else 'sharded'} "
                             f"(BASELINE config {CONFIG_OF[args.workload]} shape), "
                             f"partial_schur k={nev} max_dim={m} p={p}, "
                             f"1 step = 1 Krylov-Schur restart ({m - p} Arnoldi steps + host Schur + truncation)"),
                "n": n, "nnz_rank0": nnz_local, "nev": nev, "max_dim": m, "p": p,
                "parallelism": f"row-sharded x{world}" if world > 1 else "single GPU",
                "path": "aks_arnoldi_expand (C-chained)" if native else "python-chained stages + RCCL",
            },
            "initial_expand_ms": round(initial_ms, 2),
            "setup_s": round(t_setup, 2),
            "arnoldi_steps_timed": steps_done,
            "roofline": {
                "kernel": (("k_pb_phase1<double> + k_pb_phase2<false> (slab-binned SpMV, one pair per launch)"
                            if op.spmv_form == "binned" else "k_spmv<double,false> (CSR-stream SpMV)")
                           if world == 1 else
                           f"sharded SpMV, rank 0 (pack + all-to-all + diag[{op.spmv_form}] + off-diag)"),
                "spmv_form": op.spmv_form,
                "spmv_autotune_ms": getattr(op.diag, "tune_ms", None),
                "bound": "hbm",
                "achieved": round(achieved, 1) if achieved else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": spmv_bytes,
                "avg_launch_ms": round(spmv_avg_ms, 4),
                "launches": n_spmv,
            },
            "roofline_ortho": ortho,
            "restart_roofline": {
                "algorithmic_GB_per_restart": round(
                    ((m - p) * spmv_bytes + per_cycle + 16 * op.n_local * (m + p) + 32 * op.n_local) / 1e9, 2),
                "second_pass_fraction": round(frac_second, 3),
            },
        }
        out["restart_roofline"]["achieved_GBs"] = round(
            out["restart_roofline"]["algorithmic_GB_per_restart"] * world / (elapsed / args.steps), 1)
        out["restart_roofline"]["frac_of_peak"] = round(
            out["restart_roofline"]["achieved_GBs"] / (HBM_PEAK_GBS * world), 4)
        # SURVEY 8(d)'s own per-restart figure (three panel reads per step whether or not the second pass
        # runs): (m-p) B_spmv + 16 n 3 S(m,p) + B_tr, with S = sum of the panel widths J = p+1 .. m
        S = sum(range(p + 1, m + 1))
        survey_bytes = (m - p) * spmv_bytes + 16 * op.n_local * 3 * S + 16 * op.n_local * (m + p) + 32 * op.n_local
        out["restart_roofline"]["survey_fused_GB_per_restart"] = round(survey_bytes / 1e9, 2)
        out["restart_roofline"]["survey_fused_frac_of_peak"] = round(
            survey_bytes * world / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, n)

    # ---- extra leg: the same workload in real arithmetic (opt-in mode of the product; `value` above stays
    # the drop-in complex128 path).  Every rank takes part (collectives inside).
    real_leg = None
    if not args.no_real_leg and (world == 1 or args.real_leg):
        from arnoldi_amd.krylov_schur_real import RealKrylovSchurSolver

        del solver, ctx, op
        torch.cuda.empty_cache()
        op_r = CsrOperator(local_rows=rows, offsets=offsets, comm=comm, real=True)
        np.random.seed(0)
        rs = RealKrylovSchurSolver(op_r, nev, m, p, 1e-8, sort_key, comm=comm)
        if not native:
            rs.ctx.force_chained = True
        assert rs.start() == m
        for i in range(args.warmup):
            rs.contract(i)
            rs.expand()
        pr = None
        if native:
            pr = _hip.Probe(capacity=2 * m * args.steps + 8)
            rs.ctx.probe = pr
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            rs.contract(args.warmup + i)
            rs.expand()
        sync()
        el = time.perf_counter() - t0
        if comm is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if comm.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        real_leg = {"value": round(args.steps / el, 4), "unit": "restarts/s", "ms_per_step": round(el / args.steps * 1e3, 3),
                    "dtype": "float64 (real-packed basis, real Schur form on the host)", "spmv_form": op_r.spmv_form,
                    "spmv_autotune_ms": getattr(op_r.diag, "tune_ms", None),
                    "note": "partial_schur(arithmetic='real'): same (Q, T) contract; restart size moves by one "
                            "when it would cut a conjugate pair"}
        if pr is not None:
            ns, ms_s = pr.read(_hip.PROBE_SPMV)
            no, ms_o = pr.read(_hip.PROBE_ORTHO)
            b = op_r.algorithmic_bytes()
            real_leg.update(spmv_avg_ms=round(ms_s / max(ns, 1), 4), spmv_algorithmic_bytes=b,
                            spmv_achieved_GBs=round(b / (ms_s / max(ns, 1) * 1e-3) / 1e9, 1) if ns else None,
                            ortho_avg_ms_per_step=round(ms_o / max(no, 1), 4))
    if rank == 0:
        if real_leg is not None:
            out["real_arithmetic"] = real_leg
        print(json.dumps(out), flush=True)

    if comm is not None:
        comm.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
