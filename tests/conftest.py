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
