import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "arnoldi-py_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The package's default allocator backend is the HIP runtime alone (arnoldi_amd/mem.py).  THIS process -- the test runner --
# uses the torch interop backend: the CPU tests drive the host logic on CPU tensors (tests/fake_hip.py), the GPU tests make
# and inspect device buffers with torch.  The choice is pinned here by importing ``mem`` under the variable and restoring
# the environment afterwards, so that every child process a test starts gets the package's DEFAULT unless the test (or the
# worker script) says otherwise.  ``AKS_TEST_BACKEND=hip pytest tests/test_gpu_parity.py ...`` runs the runner itself on
# the default backend: tests that need torch buffers skip (``torch_buffers`` below), the solver-level ones run
# (tests/test_gpu_backends.py starts exactly that pass).
_had = os.environ.get("AKS_HOST_ALLOC")
os.environ["AKS_HOST_ALLOC"] = os.environ.get("AKS_TEST_BACKEND", _had or "torch")
from arnoldi_amd import mem as _mem  # noqa: E402

if _had is None:
    del os.environ["AKS_HOST_ALLOC"]
else:
    os.environ["AKS_HOST_ALLOC"] = _had
BACKEND = _mem.BACKEND


def torch_buffers():
    """For tests that hand torch device tensors to the C ABI: skip on the torch-free backend (its stream is not torch's)."""
    if BACKEND != "torch":
        pytest.skip("makes device buffers with torch: runs on the torch interop backend only")
    import torch

    return torch


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def csr_from(g, prefix):
    import scipy.sparse as sp

    shape = tuple(int(s) for s in g[prefix + "_shape"])
    return sp.csr_matrix(
        (g[prefix + "_data"], g[prefix + "_indices"], g[prefix + "_indptr"]), shape=shape
    )


@pytest.fixture
def golden():
    return load_golden
