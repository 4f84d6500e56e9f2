"""BASELINE configs 4 and 5 through the SHARDED, C-driven path at FULL size with 8 ranks on one GPU (``-m gpu``).

VERDICT r03: "the sharded code has only ever run on a 30 x 31 grid, mark(50), n = 6000 ... Eight ranks, 2M-row shards,
4M-entry ghost buffers, z-slab halos of 63 000 entries: never executed".  Here they are executed: the ranks are
threads of one worker process (tests/thread_ranks.py -- a GPU box admits six processes on its card), each driving
``aks_arnoldi_expand`` on its own shard with the ghost exchange and the all-reduces issued from C over tests/mock_rccl
(order / peer / size-checking stand-in; mailboxes sized for the 73 MB messages of the 2-rank config-5 exchange).
The worker (tests/thread_ranks_worker.py) writes what it observed; the assertions are here.

A CPU test of the in-process hand-offs themselves is at the end (no GPU needed).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

MOCK_LIB = os.path.join(ROOT, "tests", "mock_rccl", "libarnoldi_hip.so")


def _worker(tmp_path, case, extra=(), timeout=840, env_extra=None, may_fail=False):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "mock_rccl")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = os.path.join(tmp_path, f"{case}.json")
    env = dict(os.environ, AKS_LIB_PATH=MOCK_LIB, AKS_GRAPH="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "AKS_SPMV_FORM"):
        env.pop(k, None)
    env.update(env_extra or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "thread_ranks_worker.py"), "--case", case, "--out", out,
                          *extra], capture_output=True, text=True, timeout=timeout, env=env)
    if may_fail and res.returncode != 0:
        return {"failed": res.returncode, "stderr": res.stderr[-4000:]}
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-4000:]
    sys.stderr.write(res.stderr[-1500:])
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):                                  # (on the GPU box: what the worker saw, for the logs)
        import shutil

        shutil.copy(out, os.path.join(keep, f"sharded_full_{case}.json"))
    with open(out) as f:
        return json.load(f)


@pytest.mark.gpu
def test_config4_laplace3d_16m_z_slabs_eight_ranks(tmp_path):
    r = _worker(tmp_path, "c4")
    assert r["n"] == 16_002_756 and r["nnz"] == 111_638_270 and r["ranks"] == 8          # SURVEY 8 size table
    plane = 251 * 252
    assert all(o % plane == 0 for o in r["offsets"]) and sum(r["n_local"]) == r["n"]      # z-slabs: whole planes
    # halo: one plane per neighbour (two for interior slabs)
    assert r["n_ghost"] == [plane] + [2 * plane] * 6 + [plane] and r["n_send"] == r["n_ghost"]
    assert r["native"], "the ranks must run the C-driven path (library communicator, exchange + all-reduces from C)"
    assert r["H_bit_equal_across_ranks"]
    assert r["orth_err"] < 1e-11, r["orth_err"]
    assert max(r["arnoldi_residuals"]) < 1e-10, r["arnoldi_residuals"]
    # every step of a Laplacian takes the second DGKS pass: found out in the first expansion, which is repeated ONCE on
    # all ranks with the third all-reduce; four collectives per step from then on (exchange + 3 all-reduces)
    assert r["lazy_redos"] == [1] * 8 and r["collectives_per_step"] == [4] * 8
    assert r["second_passes"] == [r["second_passes"][0]] * 8 and r["second_passes"][0] >= 80        # of 40 + 2 x 25 steps
    lo, hi, im = r["ritz_hull"]
    assert -12.0 < lo and hi < 0.0 and im < 1e-8
    # same start vector on one GPU: the same Krylov-Schur trajectory up to the rounding of differently cut sums
    # (H itself is only fixed up to the phases of the Schur vectors, which follow the last bits: compare its spectrum)
    assert r["leading_ritz_rel_diff_vs_one_gpu"] < 1e-8, r
    print("C4 sharded x8:", {k: r[k] for k in ("orth_err", "leading_ritz_rel_diff_vs_one_gpu", "forms", "wall_s")})


@pytest.mark.gpu
def test_config5_random_10m_planted_eight_and_two_ranks(tmp_path):
    r = _worker(tmp_path, "c5")
    one = r["one_gpu"]
    tol = float(np.sqrt(np.finfo(np.float64).eps))
    assert r["n"] == 10_000_000
    assert one["rel_max"] < 5 * tol and max(one["rel_host"]) < 5 * tol, one      # device-side and host-side evaluation
    np.testing.assert_allclose(one["rel_device"], one["rel_host"], rtol=1e-3, atol=1e-13)
    for ranks in ("8", "2"):
        s = r["sharded"][ranks]
        assert s["native"] and s["T_bit_equal_across_ranks"], (ranks, s)
        # the planted eigenvalues are found ...
        np.testing.assert_allclose(s["vals"], r["planted"][:5], atol=0.2)
        assert s["imag_max"] < 1e-6
        np.testing.assert_allclose(s["vals"], one["vals"], rtol=1e-9)
        # ... along the one-GPU solve's trajectory (same History) and to its accuracy
        assert s["hist_restarts"] == one["hist_restarts"] and s["hist_matvecs"] == one["hist_matvecs"], (s, one)
        assert s["rel_max"] <= max(1.05 * one["rel_max"], 1e-13), (ranks, s["rel_max"], one["rel_max"])
        # exchange + 2 all-reduces per step; a third all-reduce from the restart on in which a step first needed the
        # second DGKS pass (found out on all ranks together, that expansion repeated once)
        redo = s["lazy_redos"][0]
        assert s["lazy_redos"] == [redo] * int(ranks) and redo in (0, 1) and s["collectives_per_step"] == 3 + redo
    # the exchange volumes DESIGN section 4 derives: ~65 MB per rank at 8 ranks, ~73 MB at 2 (one message)
    assert all(55e6 < b < 75e6 for b in r["sharded"]["8"]["ghost_bytes_per_spmv"]), r["sharded"]["8"]["ghost_bytes_per_spmv"]
    assert all(68e6 < b < 78e6 for b in r["sharded"]["2"]["ghost_bytes_per_spmv"]), r["sharded"]["2"]["ghost_bytes_per_spmv"]
    print("C5 sharded:", {k: (v["restarts"], v["rel_max"], v["forms"]) for k, v in r["sharded"].items()}, "one GPU:",
          one["restarts"], one["rel_max"], "wall", r["wall_s"])


@pytest.mark.gpu
def test_bench_eight_ranks_full_size_line(tmp_path):
    """``bench.py``'s rank logic (measure -> headline) with 8 ranks at n = 10M: the line the driver's --gpus 8 run prints,
    with the exchange block and rank 0's per-SpMV device-time split."""
    out = _worker(tmp_path, "bench", ["--steps", "3", "--warmup", "1", "--leg-rows", "0"])
    assert out["n_gpus"] == 8 and out["config"]["n"] == 10_000_000 and out["value"] > 0
    assert "issued from C" in out["config"]["path"] and out["config"]["parallelism"] == "row-sharded x8"
    ex = out["config"]["exchange"]
    assert 55e6 < ex["ghost_bytes_received_per_spmv_rank0"] < 75e6 and ex["collectives_per_arnoldi_step"] == 3
    split = ex["spmv_device_ms_rank0"]
    assert all(split[k] is not None and split[k] > 0 for k in ("pack", "exchange", "diag_block", "ghost_wait_plus_offdiag_block"))
    assert out["data"].startswith("rehearsal")
    # what makes the first hardware run cheap to read: the reductions' device time per step on rank 0
    # (an event pair around each aks_comm_allreduce_sum) and the model's prediction next to the measured value
    assert ex["allreduce_device_ms_per_step_rank0"] > 0 and ex["allreduce_calls_per_step_probed"] in (2.0, 3.0), ex
    assert ex["allreduce_path"].startswith("ncclAllReduce"), ex
    # (round 6: the model's kernel terms are MEASURED in the same invocation -- the restart on n / 8 rows on one GPU -- not copied
    # from an older record)
    shard = out["legs"]["one_gpu_shard"]
    assert shard["n"] == 1_250_000 and shard["spmv_avg_ms"] > 0 and shard["ortho_avg_ms_per_step"] > 0, shard
    model = out["prediction_model"]
    assert model["one_gpu_terms"].startswith("measured") and model["allreduce_us_source"] == "assumed", model
    assert abs(model["kernels_ms_per_step"] - (shard["spmv_avg_ms"] + shard["ortho_avg_ms_per_step"])) < 1e-3, model
    assert 100 < out["predicted_restarts_per_s"] < 320 and model["reductions_per_step"] == 2, model
    assert abs(out["prediction_model"]["exchange_ms_per_spmv"] - ex["ghost_bytes_received_per_spmv_rank0"] / (50e9 * 7) * 1e3) < 1e-3
    assert out["roofline"]["launches"] == 3 * 9 and out["roofline_ortho"]["launch_groups"] == 3 * 10     # (a restart's first product is the look-ahead one)
    # ... and the sharded legs of N > 1 at THEIR full sizes: Markov n = 10M (ghosts: a few grid lines), the 3-D Laplacian of
    # config 4 in z-slabs (one 252 x 253 plane per neighbour), the headline matrix real-packed (8 instead of 16 bytes per entry)
    legs = {leg["name"]: leg for leg in out["workloads"]}
    assert set(legs) == {"markov", "laplace3d", "random_real_packed"}
    assert all(leg["restarts_per_s"] > 0 and leg["path"].startswith("C-driven") and leg["n_gpus"] == 8 for leg in legs.values())
    assert legs["markov"]["n"] > 10_000_000 and legs["markov"]["exchange"]["ghost_bytes_received_per_spmv_rank0"] < 200_000
    assert legs["laplace3d"]["n"] == 252 * 253 * 254 and legs["laplace3d"]["exchange"]["ghost_bytes_received_per_spmv_rank0"] == 16 * 252 * 253
    assert legs["laplace3d"]["exchange"]["collectives_per_arnoldi_step"] == 4 and legs["laplace3d"]["second_pass_fraction"] > 0.9
    assert legs["random_real_packed"]["exchange"]["ghost_bytes_received_per_spmv_rank0"] * 2 == ex["ghost_bytes_received_per_spmv_rank0"]
    print("bench x8 rehearsal:", out["value"], "restarts/s;", split, {k: v["restarts_per_s"] for k, v in legs.items()})


@pytest.mark.gpu
def test_sharded_full_size_solve_is_bitwise_reproducible(tmp_path):
    """Two ranks x 5M rows (binned diagonal blocks: deferred normalisation, raw column carried over every restart, the
    look-ahead product of a raw column, 73 MB exchanges), the same solve three times in one process: H after every
    expansion and every contraction is the same bits each time.  Round 4 found with this that a restart's carried scale
    was lost a few times in a hundred (k_colscale_after_truncate read it through the scalar cache and cleared it with
    vector stores that could overtake the read): errors of 1e-3 .. 1e-1 in H that no small test had shown."""
    r = _worker(tmp_path, "repro", ["--ranks", "2"], timeout=400)
    assert r["forms"] == ["binned", "binned"] and r["info"][-1][3] >= 3, r["info"]      # (every re-expansion deferred)
    assert len(set(r["sha"])) == 1, (r["sha"], r["report"])
    assert all(x["first_differing_snapshot"] is None for x in r["report"]), r["report"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,args,forms,deferred", [
    ("config5_one_gpu_binned", ["--ranks", "1"], ["binned", None], True),
    ("markov_10m_one_gpu_sliced", ["--ranks", "1", "--workload", "markov"], ["sliced", None], True),
    ("config5_real_packed_one_gpu", ["--ranks", "1", "--real"], ["binned", None], True),
    ("config4_eight_ranks", ["--ranks", "8", "--workload", "laplace3d", "--rows", "16000000"], ["sliced", "csr"], None),
])
def test_full_size_drivers_are_bitwise_repeatable(tmp_path, name, args, forms, deferred):
    """The sweep round 4 ran by hand (profiles/r04_repro_sweep.txt), in the suite: each driver at a
    BASELINE size, solved twice in ONE process with a 128 MB cache sweep between the launches -- H after every expansion and
    every contraction must be the same bits both times.  Both silent wrong-H bugs of round 3 (a scalar load overtaken by
    the kernel's own stores; a last-arriver summing a stale partial) showed only at these sizes, a few times in a hundred
    restarts, and as a difference between two runs of the same solve."""
    r = _worker(tmp_path, "repro", args + ["--repeats", "2", "--sweep"], timeout=600)
    assert r["forms"][0] == forms[0] and (forms[1] is None or r["forms"][1] == forms[1]), r["forms"]
    if deferred is not None:
        assert (r["info"][-1][3] >= 3) == deferred, r["info"]        # deferred normalisation in every re-expansion
    assert len(set(r["sha"])) == 1, (r["sha"], r["report"])
    assert all(x["first_differing_snapshot"] is None and x["snapshots"] >= 8 for x in r["report"]), r["report"]
    print(name, r["n"], r["forms"], r["info"][-1], r["sha"])


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,args", [(2, []), (3, []), (4, ["--workload", "laplace3d", "--rows", "16000000"]), (8, [])])
def test_one_shot_allreduce_gives_the_collectives_bits(tmp_path, ranks, args):
    """``AKS_ALLREDUCE=oneshot``: ONE kernel per reduction writes the rank's
    [h ; ||w||^2] into a row of every peer's mailbox, polls its own arrival counter with a deadline and sums the rows in
    rank order.  Full-size sharded solves (config 5 at 2, 3 and 8 ranks, config 4 at 4: every step with the third
    reduction) must give H -- after every expansion and contraction -- bit for bit what the library collective (here: its
    order-checking stand-in, which also sums in rank order) gives, and must say that the one-shot path is the one that ran.

    Thread ranks live in ONE process, which the library votes down by default (their streams can share a hardware queue,
    see test_one_shot_allreduce_is_voted_down_between_ranks_of_one_process); the rehearsal overrides that and raises
    GPU_MAX_HW_QUEUES above its stream count.  Round 6: nothing in the exchange can block a queue any more -- a reduction
    that sat in front of its peer's post would time out and FAIL this test through ``aks_comm_status``, not hang it."""
    base = _worker(tmp_path, "repro", ["--ranks", str(ranks), "--repeats", "1"] + args, timeout=240)
    one = _worker(tmp_path, "repro", ["--ranks", str(ranks), "--repeats", "1"] + args, timeout=240,
                  env_extra={"AKS_ALLREDUCE": "oneshot", "AKS_ONESHOT_SAME_PROCESS": "1", "GPU_MAX_HW_QUEUES": "32"})
    assert base["allreduce_path"] == [0, ""] and one["allreduce_path"] == [1, ""], (base["allreduce_path"], one["allreduce_path"])
    assert one["info"] == base["info"] and one["sha"] == base["sha"], (base["sha"], one["sha"])
    print(f"one-shot x{ranks}:", one["n"], one["forms"], one["info"][-1], one["sha"])


@pytest.mark.gpu
def test_one_shot_allreduce_is_voted_down_between_ranks_of_one_process(tmp_path):
    """ADVICE r05 / VERDICT r05 item 3, in code instead of in a comment: ranks that share a process (thread ranks: the
    runtime multiplexes their streams onto GPU_MAX_HW_QUEUES hardware queues, default 4) do NOT get the one-shot exchange
    unless AKS_ONESHOT_SAME_PROCESS=1 -- ``aks_comm_create`` sees the equal pids on the ranks' cards, every rank votes the
    path down with the reason on record, and the solve runs on the library collective: same bits as without the request.
    One run at the DEFAULT queue count, no repetition, nothing that could hang."""
    small = ["--ranks", "2", "--repeats", "1", "--rows", "600000"]
    base = _worker(tmp_path, "repro", small, timeout=240)
    one = _worker(tmp_path, "repro", small, timeout=240, env_extra={"AKS_ALLREDUCE": "oneshot"})
    assert base["allreduce_path"] == [0, ""]
    assert one["allreduce_path"][0] == 0 and "ranks share a process" in one["allreduce_path"][1], one["allreduce_path"]
    assert one["info"] == base["info"] and one["sha"] == base["sha"], (base["sha"], one["sha"])


@pytest.mark.gpu
def test_one_shot_allreduce_on_a_single_hardware_queue_cannot_hang(tmp_path):
    """The deterministic form of round 5's "8 of 8 trials": thread ranks FORCED onto the exchange (override) with
    GPU_MAX_HW_QUEUES=1 -- every stream of the process on one hardware queue, the worst case of the collision round 5
    could only make unlikely.  Whatever the queue does with two ranks' kernels, the outcome is decided by construction:
    either the reductions overlap and the proof passes (path 1, the collective's bits), or a polling reduction sits in
    front of its peer's post, reaches its deadline (AKS_ONESHOT_SELFTEST_MS) and every rank votes the path down with
    "the arrival counter was not reached" (path 0, the collective's bits) -- or, if the proof passed and a LATER reduction
    collides, that reduction reaches its deadline (AKS_ONESHOT_TIMEOUT_MS) and the solve fails with the library's
    time-out message.  All three end within the time-out below; a hang -- round 5's hipStreamWaitValue64 blocked the
    queue -- is no longer a possible outcome."""
    small = ["--ranks", "2", "--repeats", "1", "--rows", "600000"]
    base = _worker(tmp_path, "repro", small, timeout=240)
    one = _worker(tmp_path, "repro", small, timeout=240, may_fail=True,
                  env_extra={"AKS_ALLREDUCE": "oneshot", "AKS_ONESHOT_SAME_PROCESS": "1", "GPU_MAX_HW_QUEUES": "1",
                             "AKS_ONESHOT_SELFTEST_MS": "500", "AKS_ONESHOT_TIMEOUT_MS": "3000"})
    if "failed" in one:
        assert "timed out after" in one["stderr"] and "one-shot all-reduce number" in one["stderr"], one["stderr"]
        print("one hardware queue: a reduction timed out inside the solve (reported, not hung)")
        return
    paths = [tuple(pw) for pw in one["allreduce_paths"]]          # (path, reason) of every rank
    if paths[0][0] == 1:
        assert all(pw == (1, "") for pw in paths), paths
    else:        # voted down TOGETHER: the rank whose reduction sat in front times out, its peer learns it from the vote
        assert all(path == 0 for path, _ in paths), paths
        assert any("arrival counter was not reached" in why for _, why in paths), paths
        assert all("arrival counter was not reached" in why or "another rank could not" in why for _, why in paths), paths
    assert one["info"] == base["info"] and one["sha"] == base["sha"], (base["sha"], one["sha"])
    print("one hardware queue:", paths)


@pytest.mark.gpu
def test_carried_scale_survives_the_truncation():
    """The restart compression with a raw last column, thousands of times: column p must inherit the scale of column m
    and every other scale must be cleared, every time (the regression test of the bug described above, on one GPU and
    without any solve around it)."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
    from arnoldi_amd import device as dev

    n, m, p = 4096, 20, 10
    basis, ws = dev.KrylovBasis(n, m), dev.Workspace(n, m)
    basis.V[:, :n].copy_(torch.randn((m + 1, n), dtype=torch.complex128, device="cuda"))
    Q = torch.eye(m, p, dtype=torch.complex128, device="cuda")
    cs = ws.buf[ws.layout.colscale_off:][: 8 * (m + 2)].view(torch.float64)
    big = torch.empty(1 << 24, dtype=torch.float64, device="cuda")          # 128 MB: sweeps the caches between launches
    bad = []
    for it in range(3000):
        cs.copy_(torch.arange(1, m + 3, dtype=torch.float64, device="cuda") * (it + 1))     # every column "raw", distinct scales
        if it % 3 == 0:
            big.fill_(float(it))
        dev.truncate(basis, m, p, Q, ws)
        got = cs.cpu().numpy()
        want = np.zeros(m + 2)
        want[p] = (m + 1) * (it + 1)                        # = the scale column m had
        want[m + 1] = (m + 2) * (it + 1)                    # (beyond the truncated range: untouched)
        if not np.array_equal(got, want):
            bad.append((it, got.tolist()))
    assert not bad, bad[:3]


@pytest.mark.gpu
def test_sequences_with_a_ghost_exchange_are_never_captured(tmp_path):
    """AKS_GRAPH=1 AKS_GRAPH_COMM=1 with a communicator in the launch sequence: operators that exchange ghost entries take
    the eager path (no hipGraph is built) and solve to the oracle's History (the guard that keeps the
    product away from the round-3 capture crash is pinned here)."""
    r = _worker(tmp_path, "graphguard", ["--ranks", "2"], timeout=400)
    for rank in r["ranks"]:
        for name, c in rank.items():
            assert c["native"] and c["any_exchange"] and c["use_graph"], (name, c)       # the switches were on ...
            assert c["graphs_built"] == 0, (name, c)                                     # ... and nothing was captured
            assert c["hist_equal"] and c["eig_err"] < 1e-9 and c["rel"] <= max(50 * c["tol"], 1e-7), (name, c)


@pytest.mark.gpu
def test_breakdown_in_the_step_that_triggers_the_lazy_redo_with_raw_columns(tmp_path):
    """Three ranks, deferred normalisation, ``on_breakdown="deflate"``, an invariant block whose six rows are spread
    over the ranks: the breakdown step is the first to need a second DGKS pass, so the expansion runs twice (the third
    all-reduce is only issued from then on) and both attempts stop among raw columns.  Complex and real-packed drivers:
    the block's wanted eigenvalues, A Q = Q T, one restart, every rank the same bits (this combination
    had no test)."""
    r = _worker(tmp_path, "breakdown", ["--ranks", "3"], timeout=400)
    rows = r["block_rows"]
    assert rows[0] < r["n"] // 3 and rows[-1] >= 2 * (r["n"] // 3 + 1), rows           # more than one rank holds the block
    for mode in ("complex", "real"):
        c = r[mode]
        assert c["T_bit_equal_across_ranks"] and c["Q_bit_equal_across_ranks"], mode
        assert c["T_diff_vs_one_gpu"] < 1e-12, (mode, c["T_diff_vs_one_gpu"])
        assert c["one_gpu"]["lazy_redos"] == 0 and c["one_gpu"]["deferred"] == 1 and c["one_gpu"]["broken"]
        for o in c["sharded"] + [c["one_gpu"]]:
            assert o["native"] and o["broken"] and o["n_iter"] == 6 and o["deferred"] == 1, (mode, o)
            assert o["restarts"] == 1 and o["hist"] == [1, 1, 1], (mode, o)
            assert o["eig_err"] < 1e-10 and o["res"] < 1e-10 and o["orth"] < 1e-12, (mode, o)
        assert all(o["lazy_redos"] == 1 for o in c["sharded"]), (mode, c["sharded"])


# ------------------------------------------------------------------------------------------------- CPU
def test_thread_comm_hand_offs():
    """The in-process stand-ins for the set-up exchanges (what torch.distributed carries between rank processes):
    all-gather, the ghost-request exchange against a brute-force answer, row gather, all-reduce, max."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from thread_ranks import run_ranks

    size, n = 5, 1000
    offs = np.linspace(0, n, size + 1).astype(np.int64)
    rng = np.random.default_rng(0)
    wanted = [np.unique(rng.integers(0, n, 200)) for _ in range(size)]
    wanted = [w[(w < offs[r]) | (w >= offs[r + 1])] for r, w in enumerate(wanted)]       # remote ids only, sorted

    def fn(comm, rank):
        counts = np.bincount(np.searchsorted(offs, wanted[rank], side="right") - 1, minlength=size)
        asked = comm.exchange_requests(wanted[rank], counts)
        ag = comm.allgather_int64([rank, rank * rank])
        rows = comm.allgather_rows(np.full((rank + 1, 2), float(rank)))
        t = torch.tensor([1.0 * rank, 2.0])
        comm.allreduce_sum_(t)
        return asked, ag, rows, t.numpy().copy(), comm.max_float(rank * 1.5)

    out = run_ranks(size, fn)
    for rank, (asked, ag, rows, t, mx) in enumerate(out):
        for peer in range(size):                                 # peer asked this rank for its ids that this rank owns
            want = wanted[peer][(wanted[peer] >= offs[rank]) & (wanted[peer] < offs[rank + 1])]
            np.testing.assert_array_equal(asked[peer], want)
        assert [list(a) for a in ag] == [[r, r * r] for r in range(size)]
        assert rows.shape == (sum(range(1, size + 1)), 2) and rows[-1, 0] == size - 1
        np.testing.assert_array_equal(t, [sum(range(size)), 2.0 * size])
        assert mx == (size - 1) * 1.5


def test_slab_offsets_cut_whole_planes():
    sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
    from arnoldi_amd.dist import slab_offsets

    offs = slab_offsets((251, 252, 253), 8)
    assert offs[0] == 0 and offs[-1] == 251 * 252 * 253 and np.all(np.diff(offs) % (251 * 252) == 0)
    assert set(np.diff(offs) // (251 * 252)) == {31, 32}
    assert list(slab_offsets((4, 3), 2)) == [0, 8, 12]
    assert list(slab_offsets((3, 2), 4)) == [0, 2, 4, 5, 6]          # fewer lines than ranks: even rows
