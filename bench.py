#!/usr/bin/env python3
"""Benchmark of the hot path: Krylov restarts / second (+ SpMV GB/s vs the HBM roofline).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "k=5 m=20", configs[4] -- fits one GPU): synthetic random CSR,
n = 10,000,000 rows, 5 non-zeros per row (nnz ~= 50M), float64 values, default_rng(1234);
nev = 5, max_dim = 20, p = 10, start vector np.random.seed(0).  A "step" is one steady-state
Krylov-Schur restart: host Schur + reorder of the 20x20 projected matrix, the truncation
V[:, :10] = V[:, :20] Qp, and 10 Arnoldi steps (SpMV + DGKS Gram-Schmidt) back to width 20.
The matrix has no dominant eigenvalues, so the solve never converges: K restarts are timed.
For N > 1 the same n = 10M problem is row-sharded over the ranks (strong scaling); without
torch.distributed.run around it, ``--gpus N`` starts its own N ranks (one process per GPU).

The ranks run on the package's DEFAULT backend -- device memory, stream and events from the HIP runtime itself, ranks
bootstrapped over ``dist.HostComm`` (TCP rendezvous next to MASTER_PORT + the library's own RCCL communicator), no torch in
the process -- i.e. on the system's ROCm.  ``AKS_HOST_ALLOC=torch`` selects the torch interop backend instead.

The line printed by rank 0 also carries
  roofline        live HIP-event timing of the SpMV launches inside the timed region,
                  algorithmic bytes (12 nnz + 36 n + 4) / average launch time vs 8 TB/s;
  roofline_ortho  the same for the Gram-Schmidt launches of the timed region;
  runtime         allocator backend, hipRuntimeGetVersion / hipDriverGetVersion, RCCL version, whether torch is in the process;
  device          sclk / mclk / power / temperature from ``rocm-smi --json`` (child process) before and after the timed region;
  calibration     GB/s of the library's plain read + write stream (``aks_stream_copy``, 50 launches) before and after, and
                  value / calibration -- comparable across boxes whose clocks differ;
  workloads       (N = 1) the other matrices of BASELINE.json -- Markov n = 10M (config 1 scaled), the 2-D Laplacian
                  of config 2, the banded and the shell-structured stand-ins for config 3 and the 3-D Laplacian of config 4 (on one GPU) --
                  through the same measurement, a few restarts each; (N > 1) Markov, 3-D Laplace and the real-packed
                  headline matrix, row-sharded over the same ranks;
  real_arithmetic (N = 1) the same default workload with partial_schur(arithmetic="real");
  cpu_baseline    (N = 1) the CPU oracle (NumPy/SciPy restatement of the reference, validated against
                  the reference's golden outputs) timed on this host at the full problem size;
  legs            (N > 1) the headline solve in the OTHER configurations, each in child processes started before the ranks
                  touch their GPUs, each under a time-out: ``oneshot`` (AKS_ALLREDUCE=oneshot), ``graph_replay`` (whole sharded
                  re-expansions replayed as hipGraphs: AKS_GRAPH=1 AKS_GRAPH_COMM=exchange), ``torch_backend`` (torch's
                  allocator + process group: its bundled HIP / RCCL), ``allreduce_probe`` (both all-reduce paths in isolation),
                  ``one_gpu_shard`` (the restart on n / N rows on one GPU: the measured terms of ``prediction_model``) --
                  restarts/s, all-reduce us per call, path taken, per-SpMV split and runtime versions per leg, and
                  ``h_agrees_with_default``: the projected matrix of the leg's first expansion against the default's (a leg
                  whose numbers came out of a broken path does not agree).
The N = 1 extra legs run in child processes after the headline measurement, so nothing in them can
cost the run its line; a failed leg is reported as {"error": ...} inside the line.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "arnoldi-py_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CONFIG_OF = {"random": 5, "laplace2d": 2, "laplace3d": 4, "markov": 1, "banded": 3, "shell": 3, "file": 3}   # BASELINE.json configs[] (1-based)
_FILE_MATRIX = {}   # --matrix: path -> CSR, loaded once per process
_RNG_LOCK = threading.Lock()   # numpy's global RNG (the reference's start-vector stream) is one per process
GPU = None     # torch.cuda.is_available(), set by run_rank: the device-timing objects need a GPU (the product
               # itself refuses to run without one; tests/bench_rehearsal.py drives the launcher / rank logic on CPU)


def parse_args(argv=None):
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
    ap.add_argument("--workload", choices=sorted(set(CONFIG_OF) - {"file"}), default="random",
                    help="random = BASELINE config 5 (default); laplace2d / laplace3d = configs 2 / 4; "
                         "markov = mark(M) of the reference's README scaled to ~n rows (sorted LR); "
                         "banded = stand-in for config 3 (af_shell10 is not available offline): use "
                         "--n 1500000 --per-row 35 --nev 20 --max-dim 41; shell = the same config with the structure of "
                         "a shell finite-element matrix (5 unknowns per node of a triangulated sheet, seven 5 x 5 "
                         "blocks per row): --n 1500000 --nev 20 --max-dim 41")
    ap.add_argument("--matrix", default=None, metavar="FILE",
                    help="a matrix file (SuiteSparse .mat with Problem.A, MatrixMarket .mtx, scipy .npz) through "
                         "arnoldi_amd.harness.load_matrix, e.g. af_shell10.mat for BASELINE config 3 (the reference's "
                         "inputs, scripts/download_matrices.sh; not obtainable offline): becomes the workload, with "
                         "--nev / --max-dim as given (config 3: --nev 20 --max-dim 41)")
    ap.add_argument("--cpu-sample-n", type=int, default=1_000_000)
    ap.add_argument("--cpu-restarts", type=int, default=3)
    ap.add_argument("--cpu-budget-s", type=float, default=150.0,
                    help="run the CPU baseline at the full size if the sample predicts at most this many seconds")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-real-leg", action="store_true",
                    help="skip the extra measurement of partial_schur(arithmetic='real') on the same matrix")
    ap.add_argument("--no-workloads", action="store_true", help="skip the Markov / Laplace legs")
    ap.add_argument("--no-device-state", action="store_true",
                    help="skip the rocm-smi samples and the streaming-copy calibration around the timed region (profiler "
                         "passes: no child process, no extra kernel in the trace)")
    ap.add_argument("--leg-rows", type=int, default=None,
                    help="(N > 1) matrix dimension of the sharded Markov / 3-D Laplace / real-packed legs instead of their "
                         "BASELINE sizes (rehearsals)")
    ap.add_argument("--chained", action="store_true",
                    help="force the Python-chained stage path on one GPU (the multi-GPU code path)")
    ap.add_argument("--arithmetic", choices=["complex", "real"], default="complex",
                    help="complex = the drop-in path (headline); real = partial_schur(arithmetic='real')")
    ap.add_argument("--probe-every", type=int, default=1,
                    help="record the HIP-event pairs around the SpMV / Gram-Schmidt launches in every K-th restart of "
                         "the timed region (1 = every restart)")
    ap.add_argument("--exchange-probe-bytes", type=int, default=8 << 20, help="(internal) bytes per peer of the exchange probe")
    ap.add_argument("--leg", choices=["measure", "cpu", "preflight", "allreduce_probe", "solve"], default=None,
                    help="(internal) run one extra leg and print its JSON object")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------- launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """``--gpus N`` without a launcher around us: start N ranks (one process per GPU), wait, forward
    rank 0's JSON line.  This parent never touches the GPU.  A failed rank => the others are stopped
    (by PID) and the exit status is non-zero."""
    port = free_port()
    rendezvous = {}
    if os.environ.get("AKS_COMM") == "host" and "AKS_RENDEZVOUS" not in os.environ:
        rendezvous["AKS_RENDEZVOUS"] = f"127.0.0.1:{free_port()}"      # the torch-free ranks' own port (dist.HostComm)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **rendezvous)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0])] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    failed = None
    line = ""
    pending = set(range(args.gpus))
    while pending and failed is None:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0:
                    failed = (r, rc)
        time.sleep(0.05)
    if failed is not None:
        for r in pending:
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    out = procs[0].stdout.read() if procs[0].stdout else ""
    for ln in out.splitlines():
        if ln.startswith("{"):
            line = ln
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with status {failed[1]}\n")
        return 1
    if not line:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------------------------- workloads
def grid_dims(workload, n):
    if workload == "markov":                      # mark(M): n = M (M + 1) / 2
        return (int(round((2 * n) ** 0.5)),)
    if workload == "laplace2d":
        nx = int(round(n ** 0.5))
        return (nx, nx + 1)
    if workload == "shell":                       # 5 unknowns per node of an nx x nx sheet
        nx = max(int(round((n / 5.0) ** 0.5)), 2)
        return (nx, nx)
    nx = int(round(n ** (1.0 / 3.0)))
    return (nx, nx + 1, nx + 2)


def file_matrix(args):
    if args.matrix not in _FILE_MATRIX:
        from arnoldi_amd import harness

        A = harness.load_matrix(args.matrix).tocsr()
        if A.shape[0] != A.shape[1]:
            raise SystemExit(f"bench.py: {args.matrix} is {A.shape[0]} x {A.shape[1]}, not square")
        _FILE_MATRIX[args.matrix] = A
    return _FILE_MATRIX[args.matrix]


def problem_size(args):
    """(n, dims), computed ONCE per process from the arguments: every rank must agree on them."""
    if args.workload == "file":
        return file_matrix(args).shape[0], None
    if args.workload in ("random", "banded"):
        return args.n, None
    dims = grid_dims(args.workload, args.n)
    if args.workload == "shell":
        return 5 * dims[0] * dims[1], dims
    n = dims[0] * (dims[0] + 1) // 2 if args.workload == "markov" else int(np.prod(dims))
    return n, dims


def build_rows(args, r0, r1, n, dims):
    from arnoldi_amd import matrices

    if args.workload == "file":
        return file_matrix(args)[r0:r1]
    if args.workload == "random":
        return matrices.random_csr(n, args.per_row, 1234, row_range=(r0, r1))
    if args.workload == "banded":
        return matrices.banded_csr(n, args.per_row, 1234)[r0:r1]
    if args.workload == "shell":
        return matrices.shell_csr(dims[0], dims[1], 5, 1234)[r0:r1]
    if args.workload == "markov":
        rows = matrices.mark(dims[0])[r0:r1]      # every rank builds the chain and keeps its rows
    else:
        rows = matrices.laplace_rows(dims, r0, r1)
    assert rows.shape == (r1 - r0, n)
    return rows


def ortho_algorithmic_bytes(n_local, J, second, deferred=False):
    """Bytes the fused Gram-Schmidt schedule must move for one step at panel width J:
    panel reads (projection, update+re-projection, second update if run) and the w traffic
    of each stage (fused figure).  ``deferred``: the expansion left its new columns raw
    (deferred normalisation), so the ``w /= beta`` pass (one read + one write of w) does not exist and
    is not counted -- a fraction must not be quoted against bytes the kernels never have to move."""
    panel = 16 * n_local * J * (3 if second else 2)
    w = 16 * n_local * (1 + 2 + (2 if second else 0) + (0 if deferred else 2))
    return panel + w


def source_stamp():
    """sha256 over the kernel source and the ABI header: profiles/pmc_summary.json carries the stamp of the
    build its counters were collected on, so a stale file is detectable."""
    h = hashlib.sha256()
    for rel in ("arnoldi-py_amd/csrc/aks_kernels.hip", "include/arnoldi_hip.h"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(args):
    """Steady-state restarts/s of the CPU oracle on this host.  A bounded sample (n = --cpu-sample-n) is timed
    first; if it predicts that the full-size problem fits --cpu-budget-s, the full size is timed and reported
    (``n == config.n``), otherwise the sample's rate scaled by the size ratio."""
    import oracle
    from arnoldi_amd import matrices

    n_full, _ = problem_size(args)

    def build(ns):
        if args.workload == "file":
            return file_matrix(args), f"{os.path.basename(args.matrix)} (n={n_full})"
        if args.workload == "random":
            return matrices.random_csr(ns, args.per_row, 1234), f"random CSR n={ns} ({args.per_row}/row, same generator)"
        if args.workload == "banded":
            return matrices.banded_csr(ns, args.per_row, 1234), f"banded CSR n={ns} ({args.per_row}/row, same generator)"
        if args.workload == "markov":
            mm = grid_dims("markov", ns)[0]
            return matrices.mark(mm), f"mark({mm})"
        if args.workload == "shell":
            sx = grid_dims("shell", ns)[0]
            return matrices.shell_csr(sx, sx, 5, 1234), f"shell CSR {sx} x {sx} nodes x 5 unknowns (same generator)"
        dims = grid_dims(args.workload, ns)
        return matrices.laplace_rows(dims, 0, int(np.prod(dims))), f"{args.workload} grid {dims}"

    def timed(ns, restarts):
        A, what = build(ns)
        A = A.astype(np.complex128)  # as the reference's scripts do (benchmark-partial-schur.py:78)
        np.random.seed(0)
        trace = {}
        t0 = time.perf_counter()
        try:
            oracle.krylov_schur(A, args.nev, max_dim=args.max_dim, max_restarts=restarts, trace=trace,
                                sort_function=oracle.arg_largest_real if args.workload == "markov" else None)
        except ValueError:
            pass
        wall = time.perf_counter() - t0
        ts = trace.get("t_restart", [])
        per = (ts[-1] - ts[0]) / (len(ts) - 1) if len(ts) >= 2 else None
        return A, what, per, wall, len(ts) - 1

    try:
        from threadpoolctl import threadpool_info

        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = len(os.sched_getaffinity(0))
    ns = n_full if args.workload == "file" else min(args.cpu_sample_n, n_full)     # (a file cannot be sampled)
    A, what, per, wall, intervals = timed(ns, 2 if ns < n_full else args.cpu_restarts)
    if per is None:
        return None
    n_used, scaled = A.shape[0], True
    sample_note = f"sample n={n_used}: {per:.3f} s per restart"
    m, p_ = args.max_dim, min(args.nev + 5, args.max_dim - 1)
    predicted = per * (n_full / n_used) * (m / (m - p_) + args.cpu_restarts)      # initial expansion + restarts
    if n_used < n_full and predicted <= args.cpu_budget_s:
        del A
        A, what, per_full, wall, intervals = timed(n_full, args.cpu_restarts)
        if per_full is not None:
            per, n_used, scaled = per_full, A.shape[0], False
    # per-phase split of the CPU path at the measured size
    rng = np.random.default_rng(0)
    xs = (rng.standard_normal(n_used) + 1j * rng.standard_normal(n_used)).astype(np.complex128)
    t0 = time.perf_counter()
    for _ in range(2):
        oracle.csr_matvec(A, xs)
    ms_matvec = (time.perf_counter() - t0) / 2 * 1e3
    scale = n_used / n_full if scaled else 1.0
    return {
        "value": (1.0 / per) * scale,
        "unit": "restarts/s",
        "cores": int(blas_threads),
        "n": int(n_used),
        "full_size": not scaled,
        "ms_per_matvec": round(ms_matvec, 2),
        "kind": "port",
        "sample": (f"oracle.krylov_schur on {what}, A.astype(complex128), {intervals} steady-state restart intervals "
                   f"timed ({per:.3f} s each, {wall:.1f} s CPU wall in all)"
                   + (f"; all work is O(n), so the rate is scaled by {n_used}/{n_full}" if scaled else
                      f" at the full problem size ({sample_note})")
                   + f"; SciPy SpMV is single-threaded, BLAS uses {blas_threads} threads of {os.cpu_count()} host CPUs"),
        "measured_restart_s": per,
        **reference_run_note(args),
    }


def reference_run_note(args):
    """The reference ITSELF cannot travel to this box; what it took where it could run -- the build container's 8 vCPUs --
    on the planted variant of the headline matrix is on record in tests/golden/g11_c5_full.npz (the fixture the GPU parity
    test solves against) and is quoted next to the port's timing, as context: it is not a measurement of this machine."""
    if args.workload != "random" or args.matrix is not None:
        return {}
    path = os.path.join(ROOT, "tests", "golden", "g11_c5_full.npz")
    try:
        g = np.load(path)
        m, k = int(g["max_dim"]), int(g["nev"])
        steps = m + (int(g["restarts"]) - 1) * (m - min(k + 5, m - 1))
        return {"reference_run_elsewhere": {
            "what": "cournape/arnoldi-py partial_schur itself, random CSR n=10M + 6 planted, k=5 m=20, to convergence",
            "restarts": int(g["restarts"]), "arnoldi_steps": steps, "wall_s": round(float(g["ref_wall_s"]), 1),
            "cores": int(g["ref_cores"]), "where": "build container (not this host)", "fixture": "tests/golden/g11_c5_full.npz"}}
    except Exception:                                      # noqa: BLE001  (context only)
        return {}


# ------------------------------------------------------------------------------------------- one measurement
def measure(args, comm, world, rank):
    """The restarts run as ``partial_schur`` runs them: host BLAS limited to one thread for the m x m Schur step
    (arnoldi_amd.utils.host_blas_threads; the CPU baseline is a process of its own and keeps its threads)."""
    from arnoldi_amd.utils import host_blas_threads

    with host_blas_threads():
        return _measure(args, comm, world, rank)


def _measure(args, comm, world, rank):
    """Build the operator for args.workload, run warmup + steps restarts, return (dict, context for extras)."""
    from arnoldi_amd import _hip, mem                      # (mem: torch's or the HIP runtime's synchronize, by backend)
    from arnoldi_amd.dist import row_offsets, slab_offsets
    from arnoldi_amd.engine import CsrOperator
    from arnoldi_amd.krylov_schur import KrylovSchurSolver
    from arnoldi_amd.utils import arg_largest_magnitude, arg_largest_real, rand_normalized_vector

    real = args.arithmetic == "real"
    n, dims = problem_size(args)
    # grids are cut into slabs of whole planes / lines along their last dimension ("Laplace: z-slabs":
    # one plane to exchange with each neighbour), everything else into equal row blocks
    offsets = slab_offsets(dims, world) if args.workload in ("laplace2d", "laplace3d") else row_offsets(n, world)
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    t_setup = time.perf_counter()
    rows = build_rows(args, r0, r1, n, dims)
    t_rows = time.perf_counter() - t_setup
    prof = None
    if os.environ.get("AKS_PROFILE_SETUP") and rank == 0:        # where the operator set-up spends its host time (stderr)
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    op = CsrOperator(local_rows=rows, offsets=offsets, comm=comm, real=real)
    if prof is not None:
        import io
        import pstats

        prof.disable()
        text = io.StringIO()
        pstats.Stats(prof, stream=text).sort_stats("cumulative").print_stats(30)
        sys.stderr.write(f"[bench] rows built in {t_rows:.2f} s; CsrOperator set-up on rank 0:\n" + text.getvalue()[:7000])
    del rows
    nnz_local = op.nnz

    nev, m = args.nev, args.max_dim
    p = min(nev + 5, m - 1)
    with _RNG_LOCK:                                    # the reference's start vector: np.random.seed(0), randn(n)
        np.random.seed(0)
        v0 = rand_normalized_vector(n, np.float64 if real else np.complex128)
    sort_key = arg_largest_real if args.workload == "markov" else arg_largest_magnitude
    if real:
        from arnoldi_amd.krylov_schur_real import RealKrylovSchurSolver

        solver = RealKrylovSchurSolver(op, nev, m, p, 1e-8, sort_key, comm=comm, v0=v0)
    else:
        solver = KrylovSchurSolver(op, nev, m, p, 1e-8, sort_key, comm=comm, v0=v0)
    del v0
    ctx = solver.ctx
    if args.chained:
        ctx.force_chained = True
    native = op.c_driven and not ctx.force_chained     # one C call per expansion (one GPU, or RCCL from C)
    t_setup = time.perf_counter() - t_setup

    def sync():
        if GPU:
            mem.synchronize()
        if comm is not None:
            comm.barrier()

    sync()
    t0 = time.perf_counter()
    assert solver.start() == m
    if GPU:
        mem.synchronize()
    initial_ms = (time.perf_counter() - t0) * 1e3
    deferred_initial = ctx.deferred_expansions > 0          # did the m-step expansion leave its columns raw?
    # What the m-step expansion produced, in two numbers: every configuration of one invocation (the legs of the N > 1 line)
    # starts from the same matrix and start vector, so their projected matrices agree to rounding -- a leg whose number
    # came out of a broken path shows here (the legs report restarts/s; this is what says the restarts were the same ones)
    Hm = np.asarray(solver.H[: m + 1, :m])
    h_check = {"fro": float(np.linalg.norm(Hm)), "abs_sum": float(np.abs(Hm).sum()), "finite": bool(np.isfinite(Hm).all())}

    for i in range(args.warmup):
        solver.contract(i)
        solver.expand()

    probe = None
    if native and GPU:
        probe = _hip.Probe(capacity=12 * m * args.steps + 8)
        ctx.probe = probe
    elif GPU:
        ctx.spmv_events = []
    second0 = int(ctx.last_ctrl.second_passes) - ctx.discarded_second_passes      # (an expansion that had to be
    steps0 = int(ctx.last_ctrl.steps_done) - ctx.discarded_steps                  # repeated is counted once)
    deferred0 = ctx.deferred_expansions

    # one mode for the whole timed region: a probed restart launches kernel by kernel (an event pair per kernel group),
    # so the un-probed restarts of --probe-every K > 1 must not replay a hipGraph in between
    graph_was = ctx.use_graph
    if probe is not None:
        ctx.use_graph = False
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        solver.contract(args.warmup + i)
        if probe is not None:
            ctx.probe = probe if i % max(args.probe_every, 1) == 0 else None
        solver.expand()
    sync()
    elapsed = time.perf_counter() - t0
    ctx.use_graph = graph_was

    if comm is not None:
        elapsed = comm.max_float(elapsed)              # the slowest rank's clock

    # ---- per-kernel device time measured inside the timed region --------------------------
    if probe is not None:
        n_spmv, spmv_ms = probe.read(_hip.PROBE_SPMV)
        n_ortho, ortho_ms = probe.read(_hip.PROBE_ORTHO)
    else:
        ev = ctx.spmv_events or []
        n_spmv, spmv_ms = len(ev), sum(a.elapsed_time(b) for a, b in ev)
        n_ortho, ortho_ms = 0, 0.0
    spmv_avg_ms = spmv_ms / max(n_spmv, 1)
    spmv_bytes = op.algorithmic_bytes()
    achieved = spmv_bytes / (spmv_avg_ms * 1e-3) / 1e9 if n_spmv and spmv_avg_ms > 0 else None

    steps_done = int(ctx.last_ctrl.steps_done) - ctx.discarded_steps - steps0
    seconds = int(ctx.last_ctrl.second_passes) - ctx.discarded_second_passes - second0
    frac_second = seconds / max(steps_done, 1)
    # the timed re-expansions either all deferred their normalisations or none did (same shape every restart)
    deferred = (ctx.deferred_expansions - deferred0) >= args.steps
    n_panel = ctx.basis.n_rows                      # rows of the panel the Gram-Schmidt kernels see
    per_cycle = 0.0   # Gram-Schmidt bytes of one restart's steps, at the measured second-pass rate
    widths = range(p + 1, m + 1)
    for J in widths:
        per_cycle += (frac_second * ortho_algorithmic_bytes(n_panel, J, True, deferred)
                      + (1 - frac_second) * ortho_algorithmic_bytes(n_panel, J, False, deferred))
    ortho = None
    if n_ortho:
        total = per_cycle * n_ortho / max(len(widths), 1)
        a = total / (ortho_ms * 1e-3) / 1e9
        ortho = {"bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(a / HBM_PEAK_GBS, 4), "launch_groups": n_ortho,
                 "avg_ms_per_step": round(ortho_ms / n_ortho, 4),
                 "second_pass_fraction": round(frac_second, 3), "normalisation_deferred": bool(deferred),
                 "traffic": None}

    # Small problems (kernels of 20-60 us) follow the host's launch latency: the same restarts again with the
    # re-expansion replayed as a hipGraph (AKS_GRAPH=1 of the product: one launch per restart), no probe.
    # With a communicator only where the engine may capture its collectives (AKS_GRAPH_COMM, engine._comm_capturable: the
    # ghost exchange on a HIP runtime >= 7.2 only); AKS_GRAPH=1 asks for it at any shard size (the "graph_replay" leg).
    graph_rate = None
    comm_in_graph = comm is not None and native and ctx._comm_capturable()
    if native and GPU and (comm is None or comm_in_graph) and (op.n_local <= 4_000_000 or os.environ.get("AKS_GRAPH") == "1"):
        ctx.probe = None
        was_graph, ctx.use_graph = ctx.use_graph, True
        for i in range(2):
            solver.contract(args.warmup + args.steps + i)
            solver.expand()
        mem.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            solver.contract(args.warmup + args.steps + 2 + i)
            solver.expand()
        mem.synchronize()
        graph_rate = args.steps / (time.perf_counter() - t0)
        ctx.use_graph = was_graph

    exchange = None
    if world > 1 or (comm is not None and comm.active):
        w = 8 if real else 16
        exchange = {"ghost_bytes_received_per_spmv_rank0": int(op.n_ghost) * w,
                    "packed_bytes_sent_per_spmv_rank0": int(getattr(op, "n_send", 0)) * w,
                    "collectives_per_arnoldi_step": ctx.collectives_per_step(),
                    "allreduce_payload_bytes": 16 * (m + 1), "lazy_redos": ctx.lazy_redos}
        if probe is not None:
            # where a sharded SpMV of rank 0 spends its device time (HIP events inside aks_shard_apply): packing the
            # entries other ranks need; the grouped send / recv (side stream); the diagonal block, which overlaps
            # it; then the wait for the ghost entries + the off-diagonal block
            split = {}
            for key, tag in (("pack", _hip.PROBE_PACK), ("exchange", _hip.PROBE_EXCHANGE), ("diag_block", _hip.PROBE_DIAG),
                             ("ghost_wait_plus_offdiag_block", _hip.PROBE_OFFDIAG)):
                k, ms = probe.read(tag)
                split[key] = round(ms / k, 4) if k else None
            exchange["spmv_device_ms_rank0"] = split
            # the reductions over the ranks between the Gram-Schmidt stages (an event pair around each one, rank 0)
            k, ms = probe.read(_hip.PROBE_ALLREDUCE)
            probed_steps = max(n_ortho, 1)
            exchange["allreduce_device_ms_per_step_rank0"] = round(ms / probed_steps, 4) if k else None
            exchange["allreduce_calls_per_step_probed"] = round(k / probed_steps, 2) if k else None
            exchange["allreduce_device_us_per_call_rank0"] = round(1e3 * ms / k, 2) if k else None
        if native and comm is not None and comm.active and GPU:
            import ctypes

            why = ctypes.create_string_buffer(256)
            path = _hip.load().aks_comm_allreduce_path(comm.native(), why, 256)
            exchange["allreduce_path"] = "one-shot mailbox exchange" if path == 1 else ("ncclAllReduce" + (f" ({why.value.decode()})" if why.value else ""))
    res = {
        "value": round(args.steps / elapsed, 4),
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "elapsed": elapsed,
        "n": n, "nnz_local": nnz_local, "nev": nev, "m": m, "p": p, "native": native,
        "initial_ms": initial_ms, "setup_s": t_setup, "steps_done": steps_done,
        "spmv_form": op.spmv_form, "spmv_tune_ms": getattr(op.diag, "tune_ms", None) or None,
        "spmv_tune_mode": getattr(op.diag, "tune_mode", None),
        "spmv_bytes": spmv_bytes, "spmv_avg_ms": spmv_avg_ms, "n_spmv": n_spmv, "achieved": achieved,
        "ortho": ortho, "frac_second": frac_second, "per_cycle": per_cycle, "n_panel": n_panel,
        "deferred": bool(deferred), "deferred_initial": bool(deferred_initial),
        "n_local": op.n_local, "exchange": exchange,
        "levels_per_round": getattr(getattr(op.diag, "binned", None), "levels_per_round", None),
        "lanes_per_wave_load": getattr(getattr(op.diag, "binned", None), "lanes_per_load", None),
        "graph_rate": graph_rate, "graphs_captured": len(ctx._graphs), "graph_capture_failures": ctx.graph_capture_failures,
        "h_check": h_check,
    }
    return res


# The scaling model of DESIGN section 4 for the headline matrix, so that ONE record of the driver's 1/2/4/8 sweep shows
# which term misses it.  Its one-GPU terms are MEASURED in the same invocation (round 6; they were constants copied from
# BENCH_r04 before): the "one_gpu_shard" leg runs a shard-sized problem -- n / N rows of the same generator, one GPU, the
# same restart -- on rank 0's GPU before the ranks touch theirs, the all-reduce latency comes from the same invocation's
# probe of the default path, and the rate of the exchange's transport from the same probe (a grouped send / recv to all peers at
# once, message size = the headline's): on a multi-GPU node no term of the model is assumed any more.
MODEL_LINK_GBS = 50.0          # fall-back: per direction per xGMI link, as DESIGN 4 assumed before there was a probe
MODEL_ALLREDUCE_US = 25.0      # fall-back when the invocation's probe has no number


def predicted_restarts_per_s(world, m, p, ghost_bytes_per_spmv, collectives_per_step, shard, allreduce_us=None, link_GBs=None):
    """restarts/s of the headline workload on ``world`` GPUs by DESIGN section 4's model.  ``shard``: the one-GPU leg of this
    invocation on n / world rows (``spmv_avg_ms``, ``ortho_avg_ms_per_step``, ``ms_per_step``)."""
    steps = m - p
    ar_us = float(allreduce_us) if allreduce_us else MODEL_ALLREDUCE_US
    link = float(link_GBs) if link_GBs else MODEL_LINK_GBS
    exch_ms = ghost_bytes_per_spmv / (link * 1e9 * max(world - 1, 1)) * 1e3 if world > 1 else 0.0
    reductions = max(int(collectives_per_step) - 1, 0) if world > 1 else 0
    kernels_ms = shard["spmv_avg_ms"] + shard["ortho_avg_ms_per_step"]
    # compression + host Schur step of the shard's restart, from the EAGER restart time (a sharded expansion is launched eagerly)
    rest_ms = max(shard.get("ms_per_step_eager_probed", shard["ms_per_step"]) - steps * kernels_ms, 0.0)
    step_ms = exch_ms + kernels_ms + reductions * ar_us * 1e-3
    restart_ms = steps * step_ms + rest_ms
    return 1e3 / restart_ms, {"exchange_ms_per_spmv": round(exch_ms, 4), "kernels_ms_per_step": round(kernels_ms, 4),
                              "reductions_per_step": reductions, "compression_plus_host_ms": round(rest_ms, 4),
                              "restart_ms": round(restart_ms, 3), "link_GBs_per_direction": round(link, 2),
                              "link_rate_source": ("this invocation's exchange probe (grouped send / recv to all peers at once)"
                                                   if link_GBs else "ASSUMED (no probe in this invocation)"),
                              "allreduce_us": round(ar_us, 2),
                              "allreduce_us_source": "this invocation's probe (ncclAllReduce)" if allreduce_us else "assumed",
                              "one_gpu_terms": "measured in this invocation (one_gpu_shard leg)"}


def spmv_kernel_name(res, world):
    if world > 1:
        return f"sharded SpMV, rank 0 (pack + exchange + diag[{res['spmv_form']}] + off-diag)"
    if res["spmv_form"] == "binned":
        return "k_pb_phase1 + k_pb_phase2 (tile-binned SpMV, one pair per launch)"
    if res["spmv_form"] == "sliced":
        return "k_sell (sliced SpMV)"
    return "k_spmv (CSR-stream SpMV)"


def pmc_traffic(res, args, world=1):
    """HBM bytes per SpMV from the committed rocprofv3 --pmc passes of this same command (profiles/collect_pmc.sh ->
    profiles/pmc_summary.json for the default workload, profiles/pmc_summary_<workload>.json for the others), valid
    only for the build they were collected on (source stamp) and the sizes they were collected at.  FETCH_SIZE is
    doubled for coalesced streams (gfx950 counts their 128-B requests as 64 B, MI355X_MICROARCH.md "HBM"): the binned
    and sliced kernels; the CSR kernel's 16-B gathers are reported raw."""
    shape = {"random": (10_000_000, 5), "markov": (10_000_000, None), "laplace2d": (1_000_000, None),
             "banded": (1_500_000, 35), "shell": (1_507_005, None), "laplace3d": (16_000_000, None)}.get(args.workload)
    if shape is None or world != 1 or args.arithmetic != "complex" or args.n != shape[0] or \
            (shape[1] is not None and args.per_row != shape[1]):
        return None, None
    name = "pmc_summary.json" if args.workload == "random" else f"pmc_summary_{args.workload}.json"
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    try:
        pmc = json.load(open(path))
        stamp = pmc.get("_source_stamp")
        if stamp != source_stamp():
            return None, f"profiles/{name} was collected on another build (stamp {stamp}); not reported"
        if res["spmv_form"] == "binned":
            return (pmc["k_pb_phase1"]["hbm_bytes_per_launch_fetch_x2"]
                    + pmc["k_pb_phase2"]["hbm_bytes_per_launch_fetch_x2"]), "rocprofv3 --pmc, same build"
        if res["spmv_form"] == "sliced":
            return pmc["k_sell"]["hbm_bytes_per_launch_fetch_x2"], "rocprofv3 --pmc, same build"
        return pmc["k_spmv"]["hbm_bytes_per_launch_raw"], "rocprofv3 --pmc, same build"
    except Exception as e:  # noqa: BLE001
        return None, f"pmc summary unreadable: {e}"


GS_FAMILIES = ("k_proj", "k_update_proj", "k_update_proj_split", "k_update", "k_reduce", "k_finish")


def pmc_ortho_traffic(res, args, world=1):
    """HBM bytes of one Gram-Schmidt step (this run's mix of panel widths) from the same committed counter passes as
    ``pmc_traffic``.  The profiled command runs the initial expansion (J = 1..m) and a few re-expansions (J = p+1..m);
    the counters are summed per kernel family over ALL of them, so what the file gives is the ratio of counted to
    algorithmic bytes over that sequence; the ratio is applied to this run's algorithmic bytes per step.  Returns
    (bytes, ratio, note)."""
    if pmc_traffic(res, args, world)[0] is None or not res["ortho"]:
        return None, None, None
    name = "pmc_summary.json" if args.workload == "random" else f"pmc_summary_{args.workload}.json"
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
        m, p, n_panel, fs = res["m"], res["p"], res["n_panel"], res["frac_second"]
        # one second-pass launch (k_update<true>, usually an immediate exit) per Gram-Schmidt step in every schedule;
        # k_finish is no longer launched when normalisation is deferred (the second-pass kernel books the step)
        steps = pmc["k_update"]["launches"]
        if steps < m or (steps - m) % (m - p):
            return None, None, f"profiles/{name}: {steps} steps are not an initial expansion plus whole re-expansions"
        re_exp = (steps - m) // (m - p)

        def alg(J, deferred):
            return (fs * ortho_algorithmic_bytes(n_panel, J, True, deferred)
                    + (1 - fs) * ortho_algorithmic_bytes(n_panel, J, False, deferred))

        # (the profiled command is this command: its initial expansion and its re-expansions defer as this run's did)
        expected = (sum(alg(J, res["deferred_initial"]) for J in range(1, m + 1))
                    + re_exp * sum(alg(J, res["deferred"]) for J in range(p + 1, m + 1)))
        counted = sum(pmc[f]["launches"] * pmc[f]["hbm_bytes_per_launch_fetch_x2"] for f in GS_FAMILIES if f in pmc)
        ratio = counted / expected
        per_step = res["per_cycle"] / max(m - p, 1)
        return int(ratio * per_step), round(ratio, 4), (
            f"rocprofv3 --pmc, same build: {counted / 1e9:.2f} GB counted over the {steps} Gram-Schmidt steps of the "
            f"profiled command against {expected / 1e9:.2f} GB algorithmic (ratio {ratio:.3f}), applied to this run's "
            f"algorithmic bytes per step")
    except Exception as e:  # noqa: BLE001
        return None, None, f"pmc summary unreadable: {e}"


def leg_summary(res, args, world=1):
    """What an extra leg reports back (a child process on one GPU; the ranks themselves for N > 1)."""
    traffic, note = pmc_traffic(res, args, world)
    out = {"spmv_traffic_bytes": traffic, "spmv_traffic_source": note, "restarts_per_s": res["value"], "ms_per_step": res["ms_per_step"], "n": res["n"], "nnz": res["nnz_local"],
           "nev": res["nev"], "max_dim": res["m"], "spmv_form": res["spmv_form"],
           "spmv_avg_ms": round(res["spmv_avg_ms"], 4), "spmv_algorithmic_bytes": res["spmv_bytes"],
           "spmv_achieved_GBs": round(res["achieved"], 1) if res["achieved"] else None,
           "spmv_frac": round(res["achieved"] / HBM_PEAK_GBS, 4) if res["achieved"] else None,
           "ortho_achieved_GBs": res["ortho"]["achieved"] if res["ortho"] else None,
           "ortho_frac": res["ortho"]["frac"] if res["ortho"] else None,
           "ortho_avg_ms_per_step": res["ortho"]["avg_ms_per_step"] if res["ortho"] else None,
           "ms_per_step_eager_probed": res["ms_per_step"],
           "second_pass_fraction": round(res["frac_second"], 3), "setup_s": round(res["setup_s"], 2)}
    o_traffic, o_ratio, _ = pmc_ortho_traffic(res, args, world)
    if o_traffic is not None:
        out["ortho_traffic_bytes_per_step"], out["ortho_traffic_over_algorithmic"] = o_traffic, o_ratio
    if res["graph_rate"] is not None:
        # Shards of <= 4M rows: the product's default (AKS_GRAPH=auto) replays the re-expansion as a hipGraph, one
        # launch per restart; that is the leg's rate.  The probed pass above has to launch kernel by kernel (a HIP
        # event pair per kernel group), so at 20-60 us per kernel it follows the box's launch latency.
        out["restarts_per_s_eager_probed"] = out["restarts_per_s"]
        out["restarts_per_s"] = out["restarts_per_s_hipgraph"] = round(res["graph_rate"], 4)
        out["ms_per_step"] = round(1e3 / res["graph_rate"], 3)
    return out


def run_child(extra_argv, timeout_s):
    """Run this script again as a child process for one extra leg; returns its JSON object or {"error": ...}."""
    cmd = [sys.executable, os.path.abspath(__file__)] + extra_argv
    try:
        cp = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout_s:.0f} s"}
    for ln in reversed(cp.stdout.splitlines()):
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                break
    tail = (cp.stderr or cp.stdout or "").strip().splitlines()[-3:]
    return {"error": f"exit status {cp.returncode}: " + " | ".join(tail)}


# ------------------------------------------------------------------------------------------- the ranks' two layers
def backend_kind():
    """"hip" (the package's default: HIP runtime allocator, ``dist.HostComm`` between ranks, no torch in the process) or
    "torch" (``AKS_HOST_ALLOC=torch``: torch allocator and streams, a torch.distributed process group between ranks)."""
    return os.environ.get("AKS_HOST_ALLOC") or "hip"


class Ranks:
    """This process's place among the ranks, on either backend: ``comm`` (None on one GPU without AKS_FORCE_COMM), and how
    to leave again.  Creating it on the hip backend touches NO GPU until ``attach_gpu()`` -- the pre-GPU legs run in
    between."""

    def __init__(self, args):
        self.kind = backend_kind()
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world} (start the ranks with "
                             f"torch.distributed.run --nproc-per-node {args.gpus}, or drop WORLD_SIZE and let "
                             f"`bench.py --gpus {args.gpus}` start them)")
        self.forced = os.environ.get("AKS_FORCE_COMM") == "1" and self.world == 1     # a one-rank communicator whose collectives run
        self.comm = None
        self.gloo = os.environ.get("AKS_BENCH_BACKEND", "nccl") != "nccl"             # torch kind: ranks share the visible GPU(s)
        if self.kind == "hip" and (self.world > 1 or self.forced):
            from arnoldi_amd.dist import HostComm

            self.comm = HostComm(rank=self.rank, size=self.world, force=self.forced) if self.forced else HostComm()

    def attach_gpu(self):
        global GPU
        from arnoldi_amd import mem

        if self.kind == "hip":
            GPU = mem.gpu_available()
            if GPU:
                mem.set_device(self.local_rank % max(mem.device_count(), 1))
            return
        import torch
        import torch.distributed as dist
        from arnoldi_amd.dist import Comm

        GPU = torch.cuda.is_available()
        local = self.local_rank % max(torch.cuda.device_count(), 1) if self.gloo else self.local_rank
        if GPU:
            torch.cuda.set_device(local)
        if self.world > 1:
            if self.gloo:
                dist.init_process_group(os.environ.get("AKS_BENCH_BACKEND"))
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            self.comm = Comm()
        elif self.forced:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
            self.comm = Comm(force=True)

    def close(self):
        if self.comm is not None:
            self.comm.barrier()
            self.comm.close()                      # the library's own RCCL communicator, if one was made (graphs first)
            if self.kind == "torch":
                import torch.distributed as dist

                dist.destroy_process_group()
            self.comm = None

    def release_device_memory(self):
        import gc

        gc.collect()
        if self.kind == "torch" and GPU:
            sys.modules["torch"].cuda.empty_cache()

    def all_ok(self, ok):
        """Whether ``ok`` holds on every rank (one small exchange; True alone on one rank)."""
        if self.comm is None or self.world == 1:
            return bool(ok)
        return all(int(v[0]) for v in self.comm.allgather_int64([1 if ok else 0]))

    def describe(self):
        from arnoldi_amd import mem

        if self.kind == "hip":
            return "dist.HostComm (TCP rendezvous + aks_comm_alltoallv), AKS_HOST_ALLOC=" + mem.BACKEND
        return "torch.distributed (" + ("gloo, ranks share the GPU" if self.gloo else "nccl") + "), AKS_HOST_ALLOC=" + mem.BACKEND


def runtime_block():
    """What the numbers of this process ran on: allocator backend, HIP runtime / driver, RCCL (None while none is loaded)."""
    from arnoldi_amd import _hip, mem

    out = {"backend": mem.BACKEND, "torch_in_process": "torch" in sys.modules}
    try:
        out.update(_hip.runtime_versions())
    except Exception as e:                                   # noqa: BLE001  (a stand-in library without the entry: reported, not fatal)
        out["versions_error"] = str(e)[:120]
    return out


# ------------------------------------------------------------------------------------------- device state + calibration
def device_telemetry(index=0, timeout_s=20):
    """Clocks, power and temperature of GPU ``index`` from ``rocm-smi --json`` in a child process -- sampled once before and
    once after the timed region, never inside it -- so that a +-4 % move of the headline between two driver records can
    be attributed (r04 -> r05 moved by -3.6 % with nothing in the record to say whether it was the box)."""
    cmd = ["rocm-smi", "-d", str(index), "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"]
    try:
        cp = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        data = json.loads(cp.stdout[cp.stdout.index("{"):])
        card = data.get(f"card{index}") or next(iter(data.values()))
    except Exception as e:                                   # noqa: BLE001  (no rocm-smi, no permission: reported, not fatal)
        return {"error": f"{type(e).__name__}: {str(e)[:100]}"}

    def pick(*needles, number=True):
        for k, v in card.items():
            if all(n in k.lower() for n in needles):
                if not number:
                    return v
                digits = "".join(ch for ch in str(v).replace("Mhz", "").replace("MHz", "") if ch.isdigit() or ch == ".")
                try:
                    return float(digits)
                except ValueError:
                    return v
        return None

    return {"sclk_mhz": pick("sclk", "clock"), "mclk_mhz": pick("mclk", "clock"), "fclk_mhz": pick("fclk", "clock"),
            "power_w": pick("power", "(w)") if pick("power", "(w)") is not None else pick("power"),
            "temp_c": pick("temperature", "junction") if pick("temperature", "junction") is not None else pick("temperature"),
            "perf_level": pick("performance level", number=False)}


def stream_copy_calibration_or_none(*a, **k):
    """The calibration is an aid: whatever goes wrong in it (no memory left, an older library without the kernel) must not
    cost the run its line."""
    try:
        return stream_copy_calibration(*a, **k)
    except Exception as e:                                   # noqa: BLE001
        sys.stderr.write(f"bench.py: streaming-copy calibration failed ({type(e).__name__}: {str(e)[:120]})\n")
        return None


def stream_copy_calibration(launches=50, mib=256):
    """GB/s of the library's plain non-temporal read + write stream (``aks_stream_copy``: 16-byte items, the access pattern
    every panel kernel is a variant of) over ``launches`` back-to-back copies of ``mib`` MiB between two device events:
    the streaming rate of THIS box, now.  ``value / calibration`` is comparable across boxes."""
    import ctypes as C

    from arnoldi_amd import _hip, mem

    nbytes = mib << 20
    device = mem.as_device(None)
    src, dst = mem.zeros(nbytes // 8, mem.f64, device), mem.empty(nbytes // 8, mem.f64, device)
    lib, stream = _hip.load(), C.c_void_p(mem.stream_ptr())
    for _ in range(3):
        _hip.check(lib.aks_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), nbytes, stream), "aks_stream_copy")
    e0, e1 = mem.Event(enable_timing=True), mem.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        lib.aks_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), nbytes, stream)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1)
    del src, dst
    return round(2.0 * nbytes * launches / (ms * 1e-3) / 1e9, 1)


# ------------------------------------------------------------------------------------------- legs of the N > 1 line
def struct_pack_double(x):
    import struct

    return struct.pack("<d", float(x))


def struct_unpack_double(b):
    import struct

    return struct.unpack("<d", b)[0]


def run_own_child(leg_argv, env, timeout_s):
    """One child process of THIS rank (``bench.py <leg_argv>``); its last JSON line, or {"error": ...}.  Started before
    the rank has touched its GPU -- a fresh child, never a re-exec of a process that has initialised the GPU."""
    proc = subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0])] + leg_argv, env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        so, se = proc.communicate(timeout=timeout_s)
        for ln in reversed(so.splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": f"exit status {proc.returncode}: " + " | ".join((se or so).strip().splitlines()[-3:])[-600:]}
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return {"error": f"timed out after {timeout_s:.0f} s"}


def preflight_child(args):
    """``--leg preflight`` (one child per rank): the C-driven collective path (aks_arnoldi_expand issuing the ghost
    exchange and the stage all-reduces on the library's communicator) against the Python-chained path on two small
    row-sharded problems -- a random graph (nearly every remote entry is exchanged) and a 2-D Laplacian (every step needs
    the second DGKS pass: the lazy third all-reduce is found out and the expansion repeated on all ranks).  On whichever
    backend the environment selects.  Prints {"ok": ..} and exits."""
    from arnoldi_amd import matrices, mem
    from arnoldi_amd.dist import row_offsets
    from arnoldi_amd.engine import CsrOperator
    from arnoldi_amd.krylov_schur import KrylovSchurSolver
    from arnoldi_amd.utils import arg_largest_magnitude

    os.environ["AKS_FORCE_COMM"] = "1" if int(os.environ.get("WORLD_SIZE", "1")) == 1 else os.environ.get("AKS_FORCE_COMM", "0")
    ranks = Ranks(args)
    ranks.attach_gpu()
    world, rank = ranks.world, ranks.rank
    out = {"ok": True, "world": world, "runtime": None}
    cases = (("random", matrices.random_csr(60_000 * world, 5, 7, planted=(4.0, 3.7, 3.4, 3.1, 2.8, 2.5)), 5, 20, 3),
             ("laplace2d", matrices.laplace2d(200, 40 * world + 1), 6, 24, 2))
    for name, A, nev, m, restarts in cases:
        n = A.shape[0]
        offs = row_offsets(n, world)
        rows = A[int(offs[rank]): int(offs[rank + 1])]
        got = {}
        for path in ("native", "python"):
            os.environ["AKS_DIST_PATH"] = path
            comm = ranks.comm                                 # (native() answers None while AKS_DIST_PATH=python)
            op = CsrOperator(local_rows=rows, offsets=offs, comm=comm)
            np.random.seed(0)
            solver = KrylovSchurSolver(op, nev, m, min(nev + 5, m - 1), 1e-10, arg_largest_magnitude, comm=comm)
            solver.start()
            Hs = [solver.H.copy()]
            for i in range(restarts):
                solver.contract(i)
                solver.expand()
                Hs.append(solver.H.copy())
            mem.synchronize()
            got[path] = (np.stack(Hs), bool(op.native_comm), int(solver.ctx.lazy_redos), int(op.n_ghost))
            del solver, op
        os.environ.pop("AKS_DIST_PATH", None)
        (Hn, nat, redo_n, ghosts), (Hp, nat_p, redo_p, _) = got["native"], got["python"]
        err = float(np.abs(Hn - Hp).max() / max(np.abs(Hp).max(), 1e-300))
        ok = nat and not nat_p and err < 1e-12 and redo_n == redo_p and np.isfinite(Hn).all()
        out[name] = {"ok": bool(ok), "max_rel_diff_H": err, "native_comm": nat, "lazy_redos": redo_n, "n_ghost_rank": ghosts}
        out["ok"] = out["ok"] and bool(ok)
    if os.environ.get("AKS_BENCH_INJECT_PREFLIGHT_FAILURE") == "1" and ranks.rank == ranks.world - 1:
        out["ok"], out["injected"] = False, "AKS_BENCH_INJECT_PREFLIGHT_FAILURE=1: what the ranks do when this check fails"
    out["ok"] = ranks.all_ok(out["ok"])
    out["runtime"] = runtime_block()
    ranks.close()
    print(json.dumps(out), flush=True)
    return 0


def allreduce_probe_child(args):
    """``--leg allreduce_probe`` (one child per rank, its own time-out): what the small reductions between the Gram-Schmidt
    stages cost on THIS machine through each of the two implementations -- ``ncclAllReduce`` and the one-shot mailbox
    kernel (AKS_ALLREDUCE=oneshot) -- 42 doubles, a batch of 200 back-to-back calls between two events, every rank.  On a
    multi-GPU node this is the number no one-GPU box can produce; a failure, a time-out or a crash here
    costs the probe, nothing else."""
    import ctypes as C

    from arnoldi_amd import _hip, mem

    os.environ["AKS_FORCE_COMM"] = "1" if int(os.environ.get("WORLD_SIZE", "1")) == 1 else os.environ.get("AKS_FORCE_COMM", "0")
    ranks = Ranks(args)
    ranks.attach_gpu()
    lib, comm, out = _hip.load(), ranks.comm, {"world": ranks.world}
    for name, env in (("nccl", None), ("oneshot", "oneshot")):
        if env is None:
            os.environ.pop("AKS_ALLREDUCE", None)
        else:
            os.environ["AKS_ALLREDUCE"] = env
        comm._destroy_native()                                # (second pass: a NEW library communicator on the same rank layer)
        handle = comm.native()
        stream = C.c_void_p(mem.stream_ptr())
        why = C.create_string_buffer(256)
        path = lib.aks_comm_allreduce_path(handle, why, 256)
        world, rank = ranks.world, ranks.rank
        buf = mem.upload(np.full(42, float(rank + 1)), mem.as_device(None))
        _hip.check(lib.aks_comm_allreduce_sum(handle, C.c_void_p(buf.data_ptr()), 42, stream), "allreduce")
        ok = bool(abs(float(np.asarray(buf.cpu().numpy())[0]) - world * (world + 1) / 2) < 1e-9)
        best = None
        for _ in range(3):
            buf.copy_(mem.host(np.full(42, 1e-300)))                # (200 sums of N equal terms stay finite)
            mem.synchronize()
            comm.barrier()
            e0, e1 = mem.Event(enable_timing=True), mem.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                lib.aks_comm_allreduce_sum(handle, C.c_void_p(buf.data_ptr()), 42, stream)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 200
            best = us if best is None else min(best, us)
        _hip.comm_status(handle)
        out[name] = {"path": "one-shot mailbox exchange" if path == 1 else "ncclAllReduce", "why_not_oneshot": why.value.decode() or None,
                     "sum_ok": ok, "device_us_per_call": round(best, 2)}
    out["slowest_rank_us_per_call"] = {k: round(ranks.comm.max_float(out[k]["device_us_per_call"]), 2) for k in ("nccl", "oneshot")}
    # ... and what the ghost exchange's transport delivers: every rank sends `per_peer` bytes to every other rank in ONE
    # grouped send / recv (aks_comm_alltoallv, the form aks_shard_apply issues), 5 back-to-back exchanges between two events.
    # per_peer ~ what a rank of the headline workload sends one peer per SpMV.  On a multi-GPU node this is the per-direction
    # link rate the scaling model otherwise has to ASSUME.
    world, rank = ranks.world, ranks.rank
    if world > 1:
        per_peer = int(min(max(args.exchange_probe_bytes, 1 << 20), 128 << 20)) & ~255
        send = mem.zeros(world * per_peer // 8, mem.f64, mem.as_device(None))
        recv = mem.empty(world * per_peer // 8, mem.f64, mem.as_device(None))
        offs = (C.c_int64 * world)(*[r * per_peer for r in range(world)])
        sizes = (C.c_int64 * world)(*[per_peer] * world)
        handle, stream = comm.native(), C.c_void_p(mem.stream_ptr())

        def exchange():
            _hip.check(lib.aks_comm_alltoallv(handle, C.c_void_p(send.data_ptr()), offs, sizes, C.c_void_p(recv.data_ptr()), offs, sizes, stream),
                       "aks_comm_alltoallv")

        exchange()
        mem.synchronize()
        comm.barrier()
        e0, e1 = mem.Event(enable_timing=True), mem.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            exchange()
        e1.record()
        e1.synchronize()
        ms = comm.max_float(e0.elapsed_time(e1) / 5)
        out["exchange_probe"] = {"bytes_per_peer": per_peer, "peers": world - 1, "slowest_rank_ms": round(ms, 4),
                                 "GBs_per_peer_per_direction": round(per_peer / (ms * 1e-3) / 1e9, 2),
                                 "what": "one grouped ncclSend / ncclRecv to and from every other rank at once (aks_comm_alltoallv)"}
    out["runtime"] = runtime_block()
    ranks.close()
    print(json.dumps(out), flush=True)
    return 0


def solve_leg_child(args):
    """``--leg solve`` (one child per rank): the HEADLINE measurement in the configuration the child's environment selects
    (AKS_ALLREDUCE, AKS_HOST_ALLOC, ...), reduced to what the legs of the N > 1 line report: restarts/s, the per-SpMV
    split, the all-reduce time per call and the path taken, and what it ran on."""
    ranks = Ranks(args)
    ranks.attach_gpu()
    res = measure(args, ranks.comm, ranks.world, ranks.rank)
    out = {"ok": True}
    if ranks.rank == 0:
        ex = res["exchange"] or {}
        out.update(restarts_per_s=res["value"], ms_per_step=res["ms_per_step"], steps=args.steps, warmup=args.warmup,
                   path="C-driven (aks_arnoldi_expand)" if res["native"] else "python-chained",
                   allreduce_path=ex.get("allreduce_path"),
                   allreduce_device_us_per_call_rank0=ex.get("allreduce_device_us_per_call_rank0"),
                   allreduce_device_ms_per_step_rank0=ex.get("allreduce_device_ms_per_step_rank0"),
                   spmv_device_ms_rank0=ex.get("spmv_device_ms_rank0"), spmv_avg_ms=round(res["spmv_avg_ms"], 4),
                   ortho_avg_ms_per_step=res["ortho"]["avg_ms_per_step"] if res["ortho"] else None,
                   restarts_per_s_hipgraph=round(res["graph_rate"], 4) if res["graph_rate"] else None,
                   graphs_captured=res["graphs_captured"], graph_capture_failures=res["graph_capture_failures"],
                   h_check=res["h_check"],
                   lazy_redos=ex.get("lazy_redos"), setup_s=round(res["setup_s"], 2), rank_layer=ranks.describe(),
                   runtime=runtime_block())
    ranks.close()
    print(json.dumps(out), flush=True)
    return 0


def ghost_bytes_per_peer(args, world):
    """Bytes a rank of the HEADLINE workload sends one peer per SpMV (uniformly random columns: a rank references
    n (N-1)/N (1 - exp(-per_row / N)) remote entries of 16 bytes, spread evenly over its N - 1 peers); 8 MiB for other workloads."""
    if world < 2 or args.workload != "random" or args.matrix is not None:
        return 8 << 20
    import math

    remote = args.n * (world - 1) / world * (1.0 - math.exp(-args.per_row / world))
    return int(16 * remote / (world - 1))


LEG_TIMEOUT_S = {"preflight": 180, "allreduce_probe": 120, "oneshot": 240, "graph_replay": 240, "torch_backend": 360, "one_gpu_shard": 240}
LEGS_BUDGET_S = 420.0          # all legs together (AKS_BENCH_LEGS_BUDGET_S), a HARD cap: a leg's time-out is cut to what is left of
                               # it, and with less than 20 s left the remaining legs are skipped -- the measurement itself must
                               # still fit the time the driver gives one bench run


def pre_gpu_legs(args, ranks):
    """N > 1 (or a forced one-rank communicator), BEFORE this process touches its GPU: every rank starts one child per leg
    -- fresh processes with their own rendezvous, each under a time-out; a leg that fails, hangs or crashes costs itself,
    nothing else -- so that the FIRST record a multi-GPU node produces decides between the configurations instead of
    measuring one of them:

      preflight       the C-driven collective path against the chained one on two small problems; if it fails on any
                      rank (all ranks learn it through the rendezvous) the measurement is handed to the torch backend,
                      see ``measure_on_the_torch_backend_instead``;
      allreduce_probe ncclAllReduce and the one-shot kernel timed in isolation, side by side;
      oneshot         the headline solve with AKS_ALLREDUCE=oneshot;
      graph_replay    the headline solve with AKS_GRAPH=1 AKS_GRAPH_COMM=exchange: after the eager (probed) restarts, the same
                      restarts with every re-expansion -- ghost exchange and reductions included -- replayed as ONE hipGraph
                      (``restarts_per_s_hipgraph``, with ``graphs_captured`` / ``graph_capture_failures``);
      torch_backend   the headline solve on the torch interop backend: torch's allocator and process group, i.e. the HIP /
                      RCCL a torch wheel bundles (7.0 / 2.26 here) instead of the system's ROCm (7.2 / 2.27);
      one_gpu_shard   rank 0 only, one GPU: the same restart on n / N rows -- the model's kernel terms, measured.

    ``value`` of the line stays the DEFAULT configuration's (this backend, ncclAllReduce), measured by the ranks themselves
    afterwards.  Returns {leg: report}."""
    hub = ranks.comm._hub
    # a rank whose child ended at once waits here for the ranks whose children run into their time-outs: the rendezvous must
    # outwait the longest leg (its default is sized for set-up exchanges), or one crashed child would cost the whole run
    patience = hub.set_timeout(max(LEG_TIMEOUT_S.values()) + 120)
    try:
        return _pre_gpu_legs(args, ranks, hub)
    finally:
        hub.set_timeout(patience)


def _pre_gpu_legs(args, ranks, hub):
    world, rank = ranks.world, ranks.rank
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    names = ["preflight", "allreduce_probe", "oneshot", "graph_replay", "torch_backend", "one_gpu_shard"]
    skip = set(filter(None, os.environ.get("AKS_BENCH_SKIP_LEGS", "").split(",")))
    ports = hub.gather(json.dumps([free_port() for _ in names]).encode() if rank == 0 else b"")[0]
    ports = dict(zip(names, json.loads(ports.decode())))
    base_env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(ranks.local_rank), WORLD_SIZE=str(world), MASTER_ADDR=host)
    for k in [k for k in base_env if k.startswith("TORCHELASTIC_") or k.startswith("TORCH_NCCL_ASYNC")] + ["AKS_DIST_PATH", "AKS_RENDEZVOUS"]:
        base_env.pop(k, None)     # under torch.distributed.run the ranks are clients of the agent's store; the children's
                                  # rank 0 hosts a store / rendezvous of its own on the fresh port
    steps, warmup = str(min(args.steps, 5)), "2"
    solve = ["--gpus", str(world), "--rows", str(args.n), "--per-row", str(args.per_row), "--nev", str(args.nev), "--max-dim",
             str(args.max_dim), "--workload", args.workload, "--steps", steps, "--warmup", warmup, "--leg", "solve"]
    if args.matrix:
        solve += ["--matrix", args.matrix]
    plan = {
        "preflight": (["--gpus", str(world), "--leg", "preflight"], {}),
        "allreduce_probe": (["--gpus", str(world), "--leg", "allreduce_probe", "--exchange-probe-bytes", str(ghost_bytes_per_peer(args, world))], {}),
        "oneshot": (solve, {"AKS_ALLREDUCE": "oneshot"}),
        "graph_replay": (solve, {"AKS_GRAPH": "1", "AKS_GRAPH_COMM": "exchange"}),
        "torch_backend": (solve, {"AKS_HOST_ALLOC": "torch"}),
    }
    out = {}
    budget, t_legs = float(os.environ.get("AKS_BENCH_LEGS_BUDGET_S", LEGS_BUDGET_S)), time.perf_counter()
    for name in names:
        if name in skip or (name == "allreduce_probe" and os.environ.get("AKS_BENCH_ALLREDUCE_PROBE", "1") == "0"):
            continue
        # rank 0's clock decides for all whether there is time for another leg (a leg that ran into its time-out -- a hang on
        # this machine -- must not be followed by four more)
        left = hub.gather(struct_pack_double(budget - (time.perf_counter() - t_legs)))[0]
        left = struct_unpack_double(left)
        if left < 20.0:
            if name != "preflight":
                out[name] = {"skipped": f"the legs' budget of {budget:.0f} s was spent", "all_ranks_ok": False}
            continue
        leg_timeout = min(float(LEG_TIMEOUT_S[name]), left)
        t0 = time.perf_counter()
        if name == "one_gpu_shard":
            if not (world > 1 and args.workload == "random" and args.matrix is None):
                continue                                      # (the model it feeds is the headline workload's at N > 1)
            report = None
            if rank == 0:                                     # one GPU, one process: rank 0's; the others wait at the gather below
                env = dict(base_env, RANK="0", LOCAL_RANK=str(ranks.local_rank), WORLD_SIZE="1")
                env.pop("AKS_FORCE_COMM", None)
                report = run_own_child(["--gpus", "1", "--rows", str(max(args.n // world, 1000)), "--per-row", str(args.per_row),
                                        "--nev", str(args.nev), "--max-dim", str(args.max_dim), "--steps", steps, "--warmup",
                                        warmup, "--leg", "measure"], env, leg_timeout)
            ok_everywhere = all(b == b"y" for b in hub.gather(b"y"))
        else:
            leg_argv, extra = plan[name]
            env = dict(base_env, MASTER_PORT=str(ports[name]), AKS_RENDEZVOUS=f"{host}:{ports[name]}", **extra)
            if world == 1:
                env["AKS_FORCE_COMM"] = "1"
            report = run_own_child(leg_argv, env, leg_timeout)
            mine_ok = "error" not in report and report.get("ok", True) is not False
            votes = [b == b"y" for b in hub.gather(b"y" if mine_ok else b"n")]
            ok_everywhere = all(votes)
            if not ok_everywhere and mine_ok:
                report = dict(report, error=f"failed on rank {votes.index(False)}")
        if report is not None:
            report["all_ranks_ok"] = bool(ok_everywhere)
            report["seconds"] = round(time.perf_counter() - t0, 1)
            out[name] = report
    return out


def compare_legs_with_default(legs, mine):
    """The solve legs against the default configuration's own measurement: the projected matrix of the first expansion
    (``h_check``: Frobenius norm and absolute sum), relative difference.  Rounding level is expected -- a collective sums in
    another order than the one-shot kernel --; anything larger, or a non-finite H, is a leg that solved something else."""
    for leg in legs.values():
        theirs = leg.get("h_check") if isinstance(leg, dict) else None
        if theirs:
            leg["h_vs_default_rel_diff"] = max(abs(theirs[k] - mine[k]) / max(abs(mine[k]), 1e-300) for k in ("fro", "abs_sum"))
            leg["h_agrees_with_default"] = bool(theirs["finite"] and mine["finite"] and leg["h_vs_default_rel_diff"] < 1e-10)
    return legs


def measure_on_the_torch_backend_instead(args, argv, ranks, legs):
    """The default configuration's collective path failed its preflight on some rank: a measurement on it would at best be the
    host-staged chained path (every ghost exchange through the TCP rendezvous: seconds per SpMV at n = 10M) -- not a number
    worth having.  The ranks have not touched their GPUs yet, so each starts ONE child that runs this same command on the
    torch interop backend -- its C-driven path if the ``torch_backend`` leg just showed it working, else the Python-chained
    stages over torch.distributed -- and rank 0 forwards that child's line, with the legs and the reason beside it.  The
    driver's first multi-GPU record must not be empty because ONE configuration does not work there."""
    pre = legs["preflight"]
    torch_leg = legs.get("torch_backend") or {}
    chained = "error" in torch_leg or not torch_leg.get("all_ranks_ok", False)
    sys.stderr.write(f"bench.py: rank {ranks.rank}: preflight of the default configuration failed ({pre.get('error') or pre}); "
                     f"measuring on the torch backend ({'Python-chained' if chained else 'C-driven'} path)\n")
    ranks.comm._hub.gather(b"leaving")               # every rank has its verdict: the rendezvous can go
    ranks.comm.close()
    env = dict(os.environ, AKS_HOST_ALLOC="torch", AKS_BENCH_PREFLIGHT="0")
    env.pop("AKS_RENDEZVOUS", None)
    if chained:
        env["AKS_DIST_PATH"] = "python"
    proc = subprocess.run([sys.executable, os.path.abspath(sys.argv[0])] + argv, env=env, stdout=subprocess.PIPE, text=True)
    if ranks.rank == 0:
        line = next((ln for ln in reversed(proc.stdout.splitlines()) if ln.startswith("{")), None)
        if line is None:
            sys.stderr.write("bench.py: the torch-backend measurement printed no line either\n")
            return proc.returncode or 1
        out = json.loads(line)
        out["legs"] = {k: v for k, v in legs.items() if k != "preflight"}
        out["config"]["native_preflight"] = pre
        out["config"]["fallback"] = ("the default configuration (HIP-runtime backend, dist.HostComm) failed its preflight; this line was "
                                     "measured on the torch interop backend, " + ("Python-chained stages" if chained else "C-driven path"))
        emit(json.dumps(out))
    return proc.returncode


# ------------------------------------------------------------------------------------------- rank main
def emit(line):
    """The ONE line of this process on the real stdout (fd 1 is pointed at stderr while the rank runs, so that
    banners of RCCL / the HIP runtime cannot get in front of it)."""
    os.write(_REAL_STDOUT, (line + "\n").encode())


_REAL_STDOUT = 1


def model_fields(res, args, world, legs=None):
    """``predicted_restarts_per_s`` next to the measured ``value`` (N > 1, headline workload): DESIGN section 4's model with
    the one-GPU terms and the all-reduce latency MEASURED in this invocation's legs."""
    ex = res.get("exchange")
    shard = (legs or {}).get("one_gpu_shard") or {}
    if world <= 1 or not ex or args.workload != "random" or args.matrix is not None or args.arithmetic != "complex" \
            or "error" in shard or not shard.get("spmv_avg_ms") or not shard.get("ortho_avg_ms_per_step"):
        return {}
    probe = (legs or {}).get("allreduce_probe") or {}
    ar_us = (probe.get("slowest_rank_us_per_call") or {}).get("nccl")
    link = (probe.get("exchange_probe") or {}).get("GBs_per_peer_per_direction")
    rate, parts = predicted_restarts_per_s(world, res["m"], res["p"], ex["ghost_bytes_received_per_spmv_rank0"],
                                           ex["collectives_per_arnoldi_step"], shard, ar_us, link)
    return {"predicted_restarts_per_s": round(rate, 2), "prediction_model": parts}


def headline(res, args, world, comm_forced=False, preflight=None):
    """The JSON object of the headline measurement (what rank 0 prints), from ``measure``'s result."""
    n, m, p, nev = res["n"], res["m"], res["p"], res["nev"]
    traffic, traffic_note = pmc_traffic(res, args, world)
    achieved, spmv_bytes = res["achieved"], res["spmv_bytes"]
    out = {
        "metric": "krylov_restarts_per_sec",
        "value": res["value"],
        "unit": "restarts/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"],
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "complex128" if args.arithmetic == "complex" else "float64",
        "data": "synthetic" if args.matrix is None else f"file {os.path.basename(args.matrix)}",
        "config": {
            "workload": ((f"{os.path.basename(args.matrix)}" if args.matrix else f"{args.workload} CSR")
                         + f" n={n} nnz={res['nnz_local'] if world == 1 else 'sharded'} "
                         f"(BASELINE config {CONFIG_OF[args.workload]} shape), "
                         f"partial_schur k={nev} max_dim={m} p={p}, "
                         f"1 step = 1 Krylov-Schur restart ({m - p} Arnoldi steps + host Schur + truncation)"),
            "n": n, "nnz_rank0": res["nnz_local"], "nev": nev, "max_dim": m, "p": p,
            "parallelism": f"row-sharded x{world}" if world > 1 else "single GPU",
            "path": (("aks_arnoldi_expand: one C call per expansion"
                      + (", RCCL all-reduces and ghost exchange issued from C" if res["exchange"] or comm_forced else ""))
                     if res["native"] else "python-chained stages + torch.distributed collectives"),
            "exchange": res["exchange"],
            "native_preflight": preflight,
        },
        "initial_expand_ms": round(res["initial_ms"], 2),
        "setup_s": round(res["setup_s"], 2),
        "arnoldi_steps_timed": res["steps_done"],
        **model_fields(res, args, world),
        "roofline": {
            "kernel": spmv_kernel_name(res, world),
            "spmv_form": res["spmv_form"],
            "spmv_form_chosen_by": res["spmv_tune_mode"],
            "spmv_autotune_ms": res["spmv_tune_ms"],
            "bound": "hbm",
            "achieved": round(achieved, 1) if achieved else None,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
            "traffic": traffic,
            "traffic_source": traffic_note,
            "algorithmic_bytes_per_launch": spmv_bytes,
            "avg_launch_ms": round(res["spmv_avg_ms"], 4),
            "launches": res["n_spmv"],
        },
        "roofline_ortho": res["ortho"],
    }
    if res["ortho"]:
        o_traffic, o_ratio, o_note = pmc_ortho_traffic(res, args, world)
        res["ortho"].update(traffic=o_traffic, traffic_over_algorithmic=o_ratio, traffic_source=o_note,
                            algorithmic_bytes_per_step=int(res["per_cycle"] / max(m - p, 1)))
    n_loc, per_cycle, elapsed = res["n_panel"], res["per_cycle"], res["elapsed"]
    rr = {"algorithmic_GB_per_restart": round(((m - p) * spmv_bytes + per_cycle + 16 * n_loc * (m + p) + 32 * n_loc) / 1e9, 2),
          "second_pass_fraction": round(res["frac_second"], 3)}
    rr["achieved_GBs"] = round(rr["algorithmic_GB_per_restart"] * world / (elapsed / args.steps), 1)
    rr["frac_of_peak"] = round(rr["achieved_GBs"] / (HBM_PEAK_GBS * world), 4)
    # SURVEY 8(d)'s own per-restart figure (three panel reads per step whether or not the second pass
    # runs): (m-p) B_spmv + 16 n 3 S(m,p) + B_tr, with S = sum of the panel widths J = p+1 .. m
    S = sum(range(p + 1, m + 1))
    survey_bytes = (m - p) * spmv_bytes + 16 * n_loc * 3 * S + 16 * n_loc * (m + p) + 32 * n_loc
    rr["survey_fused_GB_per_restart"] = round(survey_bytes / 1e9, 2)
    rr["survey_fused_frac_of_peak"] = round(survey_bytes * world / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4)
    out["restart_roofline"] = rr
    return out


def sharded_legs(args, comm, world, rank, log=None):
    """N > 1: north_star asks for restarts/s "on synthetic Markov/Laplace CSR at 1/2/4/8 GPUs", and
    the headline's uniformly random matrix is the one workload whose ghost exchange cannot shrink with N.  After the
    headline, in the SAME rank processes: the Markov chain (n = 10M, a few grid lines of ghosts), the 3-D Laplacian of
    BASELINE config 4 (n = 16M, z-slabs: one plane per neighbour) and the headline matrix in real-packed arithmetic
    (half the exchange volume) -- each with its exchange block and per-SpMV split.  ``--leg-rows R`` shrinks all three
    (rehearsals).  Collective: every rank runs every leg; a leg that fails on one rank fails on all (same inputs)."""
    import copy
    import gc

    legs = []
    small = args.leg_rows
    specs = (("markov", dict(workload="markov", n=small or 10_000_000, nev=5, max_dim=20, arithmetic="complex")),
             ("laplace3d", dict(workload="laplace3d", n=small or 16_000_000, nev=10, max_dim=40, arithmetic="complex")),
             ("random_real_packed", dict(workload="random", n=small or args.n, nev=args.nev, max_dim=args.max_dim,
                                         per_row=args.per_row, arithmetic="real")))
    for name, over in specs:
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        a.steps, a.warmup = min(args.steps, 5), 2
        gc.collect()
        if GPU and "torch" in sys.modules:                # (the torch-free ranks free their HIP allocations with the objects)
            sys.modules["torch"].cuda.empty_cache()
        t0 = time.perf_counter()
        res = measure(a, comm, world, rank)
        leg = leg_summary(res, a, world)
        leg.update(name=name, n_gpus=world, exchange=res["exchange"], wall_s=round(time.perf_counter() - t0, 1),
                   dtype="float64 (real-packed)" if over["arithmetic"] == "real" else "complex128",
                   path="C-driven (aks_arnoldi_expand)" if res["native"] else "python-chained")
        if log is not None:
            log(f"leg {name}: {leg['restarts_per_s']} restarts/s")
        legs.append(leg)
        del res
    return legs


def run_rank(args, argv):
    """One rank: (hip kind, N > 1) pre-GPU legs in child processes; then the measurement on the default configuration; the
    sharded legs (N > 1) or the child-process legs (N = 1); ONE line from rank 0."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    ranks = Ranks(args)
    world, rank = ranks.world, ranks.rank
    legs = None
    if (ranks.kind == "hip" and ranks.comm is not None and args.leg is None and "AKS_DIST_PATH" not in os.environ
            and os.environ.get("AKS_BENCH_PREFLIGHT", "1") != "0"):
        legs = pre_gpu_legs(args, ranks)
        if "preflight" in legs and not legs["preflight"]["all_ranks_ok"]:
            return measure_on_the_torch_backend_instead(args, argv, ranks, legs)
    ranks.attach_gpu()
    telemetry = calibration = None
    if rank == 0 and GPU and args.leg is None and not args.no_device_state:
        # (the calibration first: both samples of the device state are then taken right behind 50 streaming launches -- an idle
        # GPU parks at ~160 MHz, which says nothing about the clocks the timed region will run at)
        calibration = {"stream_copy_GBs_before": stream_copy_calibration_or_none()}
        telemetry = {"before": device_telemetry(ranks.local_rank)}

    res = measure(args, ranks.comm, world, rank)

    if args.leg == "measure":                      # child process of a one-GPU run: report and leave
        emit(json.dumps(leg_summary(res, args)))
        return 0

    out = None
    if rank == 0:
        out = headline(res, args, world, ranks.forced, legs.get("preflight") if legs else None)
        out["config"]["rank_layer"] = ranks.describe() if ranks.comm is not None else "single process, AKS_HOST_ALLOC=" + backend_kind()
        out["runtime"] = runtime_block()
        if GPU and calibration is not None:
            calibration["stream_copy_GBs_after"] = stream_copy_calibration_or_none()
            rates = [v for v in (calibration["stream_copy_GBs_before"], calibration["stream_copy_GBs_after"]) if v]
            calibration.update(kernel="aks_stream_copy (non-temporal 16-byte read + write stream)", launches=50,
                               bytes_moved_per_launch=2 * (256 << 20))
            if rates:
                mean = sum(rates) / len(rates)
                calibration.update(of_hbm_peak=round(mean / HBM_PEAK_GBS, 4), value_per_copy_TBs=round(res["value"] / (mean / 1e3), 3))
            telemetry["after"] = device_telemetry(ranks.local_rank)
            out["calibration"], out["device"] = calibration, telemetry
        out["h_check"] = res["h_check"]
        if legs:
            out["legs"] = compare_legs_with_default({k: v for k, v in legs.items() if k != "preflight"}, res["h_check"])
            out.update(model_fields(res, args, world, legs))

    # ---- N > 1: the workloads that can scale, through the same ranks (no child processes once the GPUs are in use)
    if world > 1 and args.workload == "random" and args.arithmetic == "complex" and not args.no_workloads:
        more = sharded_legs(args, ranks.comm, world, rank)
        if rank == 0:
            out["workloads"] = more

    ranks.close()

    # ---- extra legs (one GPU only): child processes, after this process has released the GPU memory
    if rank == 0 and world == 1 and GPU:
        del res
        ranks.release_device_memory()
        base = ["--steps", str(min(args.steps, 5)), "--warmup", "2", "--leg", "measure"]
        if args.workload == "random" and args.arithmetic == "complex":
            if not args.no_real_leg:
                leg = run_child(["--rows", str(args.n), "--per-row", str(args.per_row), "--nev", str(args.nev),
                                 "--max-dim", str(args.max_dim), "--arithmetic", "real"] + base, 600)
                if "error" not in leg:
                    leg.update(value=leg["restarts_per_s"], unit="restarts/s")
                    leg.update(dtype="float64 (real-packed basis, real Schur form on the host)",
                               note="partial_schur(arithmetic='real'): same (Q, T) contract; restart size moves by "
                                    "one when it would cut a conjugate pair")
                out["real_arithmetic"] = leg
            if not args.no_workloads:
                more = []
                for name, extra in (("markov", ["--workload", "markov", "--rows", "10000000"]),
                                    ("laplace2d", ["--workload", "laplace2d", "--rows", "1000000", "--nev", "10",
                                                   "--max-dim", "40"]),                       # BASELINE config 2
                                    ("banded", ["--workload", "banded", "--rows", "1500000", "--per-row", "35",
                                                "--nev", "20", "--max-dim", "41"]),           # config 3 stand-in
                                    ("shell", ["--workload", "shell", "--rows", "1507005", "--nev", "20",
                                               "--max-dim", "41"]),                           # config 3, FEM-shell structure
                                    ("laplace3d", ["--workload", "laplace3d", "--rows", "16000000", "--nev", "10",
                                                   "--max-dim", "40"])):                      # config 4 on ONE GPU
                    leg = run_child(extra + base, 600)
                    leg["name"] = name
                    more.append(leg)
                out["workloads"] = more
        if not args.no_cpu_baseline:
            leg = run_child((["--matrix", args.matrix] if args.matrix else []) +
                            ["--leg", "cpu", "--workload", "random" if args.matrix else args.workload, "--rows", str(args.n), "--per-row",
                             str(args.per_row), "--nev", str(args.nev), "--max-dim", str(args.max_dim),
                             "--cpu-sample-n", str(args.cpu_sample_n), "--cpu-restarts", str(args.cpu_restarts),
                             "--cpu-budget-s", str(args.cpu_budget_s)], 900)
            out["cpu_baseline"] = leg
    if rank == 0:
        out["torch_in_process"] = "torch" in sys.modules
        emit(json.dumps(out))
    return 0


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.matrix is not None:
        args.workload = "file"
    if args.leg == "cpu":
        print(json.dumps(cpu_baseline(args)), flush=True)
        return 0
    if args.leg == "preflight":
        return preflight_child(args)
    if args.leg == "allreduce_probe":
        return allreduce_probe_child(args)
    if args.leg == "solve":
        return solve_leg_child(args)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    return run_rank(args, argv)


if __name__ == "__main__":
    sys.exit(main())
