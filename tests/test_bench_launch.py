"""bench.py as the driver runs it: ``python bench.py --gpus N`` must start its own N ranks (VERDICT r01:
`assert world == args.gpus` killed it).  Rehearsed here on CPU through tests/bench_rehearsal.py, which swaps the device
entry points for tests/fake_hip.py (test infrastructure), makes the ranks talk over gloo and runs bench.main()
unchanged; bench.py itself carries no test double."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra, timeout=600):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_rehearsal.py")] + extra, capture_output=True,
                          text=True, timeout=timeout, env=env)


def test_bench_carries_no_test_double():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "fake_hip" not in src and "FAKE" not in src


def test_bench_starts_its_own_ranks():
    res = _run(["--gpus", "2", "--rows", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--leg-rows", "4000"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    # N > 1: the workloads that can scale (Markov, 3-D Laplace in z-slabs, real-packed) are measured by the same ranks
    legs = {leg["name"]: leg for leg in out["workloads"]}
    assert set(legs) == {"markov", "laplace3d", "random_real_packed"}
    assert all(leg["n_gpus"] == 2 and leg["restarts_per_s"] > 0 and leg["exchange"]["ghost_bytes_received_per_spmv_rank0"] > 0
               for leg in legs.values()), legs
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["metric"] == "krylov_restarts_per_sec"
    assert out["scaling"] == "strong" and out["config"]["parallelism"] == "row-sharded x2"
    ex = out["config"]["exchange"]
    assert ex["ghost_bytes_received_per_spmv_rank0"] > 0 and ex["collectives_per_arnoldi_step"] in (3, 4)
    assert 0 < out["config"]["nnz_rank0"] < 20000 * 5
    assert out["data"].startswith("rehearsal")              # a CPU stand-in never reports as a measurement


def test_bench_under_torch_distributed_run():
    """The driver's own launch for N > 1: ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher's environment): one line, from rank 0."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "tests", "bench_rehearsal.py"), "--gpus", "2", "--rows", "20000", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--no-workloads"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "row-sharded x2" and out["value"] > 0


def test_bench_reports_a_failed_rank():
    # nev > max_dim - 1: every rank trips the reference's assertion; the launcher must come back non-zero
    res = _run(["--gpus", "2", "--rows", "5000", "--steps", "1", "--warmup", "0", "--nev", "5", "--max-dim", "4",
                "--no-cpu-baseline"], timeout=300)
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_a_half_launched_world():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_rehearsal.py"), "--gpus", "2", "--rows", "5000"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr


def test_bench_takes_a_matrix_file(tmp_path):
    """``bench.py --matrix FILE``: a supplied file (SuiteSparse .mat / MatrixMarket / .npz through
    arnoldi_amd.harness.load_matrix) becomes the workload -- how af_shell10 (BASELINE config 3; not obtainable
    offline) would be measured."""
    import scipy.io
    import scipy.sparse as sp

    sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
    from arnoldi_amd import matrices

    A = sp.csr_matrix(matrices.banded_csr(3000, 9, 7))
    path = os.path.join(tmp_path, "band3000.mtx")
    scipy.io.mmwrite(path, A)
    res = _run(["--matrix", path, "--nev", "4", "--max-dim", "12", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["config"]["n"] == 3000 and out["config"]["nnz_rank0"] == A.nnz
    assert "band3000.mtx" in out["config"]["workload"] and out["config"]["nev"] == 4 and out["config"]["max_dim"] == 12
