"""The parity files once more on the package's DEFAULT backend (``-m gpu``).

The test runner itself uses the torch interop backend (tests/conftest.py: the GPU tests make and inspect device buffers
with torch).  The product's default is the torch-free one -- HIP runtime allocator, its own stream per host thread, the
system's ROCm -- so the solver-level parity tests (golden solves, the reference's own tests restated, the stress grid,
explicit restarts, real arithmetic, locking, deferred normalisation, look-ahead, graph replay, two solving threads) run a
second time there: ONE child pytest process with ``AKS_TEST_BACKEND=hip`` over the three parity files.  Tests that hand
torch tensors to the C ABI skip in that pass (``conftest.torch_buffers``); the full-size and multi-process cases -- which
start workers of their own, on whichever backend the worker names -- are left to the first pass."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

# left to the first pass: full-size problems (minutes), tests that only start worker processes (their backend is the worker's)
# ... and the long grids (the reference's 18-case stress grid and the larger deflation cases: a minute between them): the second
# pass keeps one representative of each family (golden solves, reference tests, tiny sizes, locking, real arithmetic, explicit
# restarts, graph replay, two threads) -- the driver runs the whole -m gpu suite in one call and its time matters
FIRST_PASS_ONLY = ("full_size or row_sharded or bench or one_shot or rccl or c_abi or without_torch or capturable "
                   "or sharded or stress_grid or deflation_against_oracle_larger")


def test_parity_files_pass_on_the_default_backend(tmp_path):
    env = dict(os.environ, AKS_TEST_BACKEND="hip")
    env.pop("AKS_HOST_ALLOC", None)
    files = [os.path.join(ROOT, "tests", f) for f in ("test_gpu_parity.py", "test_gpu_real.py", "test_gpu_explicit.py")]
    res = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                          "-k", f"not ({FIRST_PASS_ONLY})", "-rs"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = res.stdout[-6000:]
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "default_backend_pass.log"), "w") as f:
            f.write(res.stdout + res.stderr)
    assert res.returncode == 0, tail + res.stderr[-3000:]
    summary = re.search(r"(\d+) passed(?:, (\d+) skipped)?", res.stdout)
    assert summary, tail
    passed, skipped = int(summary.group(1)), int(summary.group(2) or 0)
    # most of the selection runs on either backend; what skips says why ("makes device buffers with torch")
    assert passed >= 60 and passed > skipped / 2, tail
    reasons = set(re.findall(r"SKIPPED \[\d+\] [^:]+:\d+: (.*)", res.stdout))
    assert all("torch" in r for r in reasons), reasons
    print(f"default-backend pass: {passed} passed, {skipped} skipped (torch buffers)")


def test_the_second_pass_really_ran_on_the_hip_backend():
    """Guard of the guard: with AKS_TEST_BACKEND=hip the runner's ``mem`` is the HIP backend and torch is not imported by
    importing the package (so a pass that silently ran on torch cannot be mistaken for this one)."""
    env = dict(os.environ, AKS_TEST_BACKEND="hip")
    env.pop("AKS_HOST_ALLOC", None)
    code = ("import sys; sys.path.insert(0, %r); import conftest; import arnoldi_amd; "
            "print(conftest.BACKEND, 'torch' in sys.modules, arnoldi_amd.mem.gpu_available())" % os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["hip", "False", "True"], out.stdout
