"""CPU-only tests: the C-ABI library loads and exports what include/arnoldi_hip.h declares,
the pure-host entry points, the host-side mirror of the reference interface (matrices,
ordered Schur, History, driver control flow) and the row-sharded driver over ``gloo``.

Device entry points are replaced by tests/fake_hip.py where a test needs the driver to run
end to end; the real kernels are covered by tests/test_gpu_parity.py on the MI355X.
"""
import ctypes as C
import json
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from conftest import ROOT, csr_from, load_golden

C128 = np.complex128


# ---------------------------------------------------------------------------- the boundary
def _declared_functions():
    text = open(os.path.join(ROOT, "include", "arnoldi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aks_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from arnoldi_amd import _hip

    declared = _declared_functions()
    assert len(declared) >= 18
    raw = C.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in arnoldi_hip.h but not exported"
    assert sorted(_hip.SIGNATURES) == declared, "ctypes binding and header disagree"
    lib = _hip.load()
    assert lib.aks_abi_version() == _hip.ABI_VERSION == 6
    from arnoldi_amd._version import ABI

    assert ABI == _hip.ABI_VERSION


def test_header_constants_match_binding():
    from arnoldi_amd import _hip

    text = open(os.path.join(ROOT, "include", "arnoldi_hip.h")).read()
    consts = dict(re.findall(r"#define (AKS_[A-Z_]+) (\(?-?\d+\)?)", text))
    assert int(consts["AKS_MAX_DIM"]) == _hip.MAX_DIM
    assert int(consts["AKS_MAX_TRUNC"]) == _hip.MAX_TRUNC
    assert int(consts["AKS_SPMV_TILE_NNZ"]) == _hip.SPMV_TILE_NNZ
    assert C.sizeof(_hip.Ctrl) == 64


def test_workspace_layout_and_errors():
    from arnoldi_amd import _hip

    lay = _hip.workspace_layout(10_000_000, 20)
    assert lay.n_blocks == 1024 and lay.ld_partial == 22 and lay.red_len == 22
    offs = [lay.ctrl_off, lay.red1_off, lay.red2_off, lay.red3_off, lay.partial_off, lay.total_bytes]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert lay.total_bytes >= lay.partial_off + 1024 * 22 * 16
    assert _hip.workspace_layout(100, 5).n_blocks == 1
    with pytest.raises(_hip.HipLibraryError, match="max_dim"):
        _hip.workspace_layout(100, 129)
    with pytest.raises(_hip.HipLibraryError, match="n_rows"):
        _hip.workspace_layout(0, 5)


def _plan(indptr):
    from arnoldi_amd import _hip

    indptr = np.ascontiguousarray(indptr, np.int32)
    n = indptr.size - 1
    out = np.empty(n + 2, np.int32)
    nt = _hip.load().aks_csr_plan_tiles(indptr.ctypes.data, n, _hip.SPMV_TILE_NNZ, out.ctypes.data, out.size)
    assert nt > 0
    return out[: nt + 1]


def test_tile_planner():
    # 5 per row -> 51 rows per tile (255 nnz)
    t = _plan(np.arange(0, 5 * 1000 + 1, 5))
    assert t[0] == 0 and t[-1] == 1000 and np.all(np.diff(t)[:-1] == 51)
    # ragged: every tile holds <= 256 nnz unless it is a single row; row cap on empty runs
    rng = np.random.default_rng(0)
    lengths = rng.integers(0, 40, 5000)
    lengths[7] = 1000
    lengths[2000:3500] = 0
    indptr = np.concatenate([[0], np.cumsum(lengths)])
    t = _plan(indptr)
    assert t[0] == 0 and t[-1] == 5000 and np.all(np.diff(t) > 0)
    nnz = indptr[t[1:]] - indptr[t[:-1]]
    rows = np.diff(t)
    assert np.all((nnz <= 256) | (rows == 1))
    assert rows.max() <= 4 * 256
    assert 7 in t and 8 in t  # the long row stands alone


# ---------------------------------------------------------------------------- matrices
def test_mark_matches_reference_bit_for_bit():
    from arnoldi_amd.matrices import mark

    g = load_golden("g1_matrices")
    for m in (2, 3, 10, 50):
        ref, mine = csr_from(g, f"mark{m}"), mark(m)
        np.testing.assert_array_equal(mine.indptr, ref.indptr)
        np.testing.assert_array_equal(mine.indices, ref.indices)
        np.testing.assert_array_equal(mine.data, ref.data)
    # literals of the reference's tests/test_matrices.py:10-28
    np.testing.assert_array_almost_equal(mark(2).toarray(), [[0, 1, 1], [0.5, 0, 0], [0.5, 0, 0]])
    np.testing.assert_array_almost_equal(mark(3).toarray(), [
        [0., 0.5, 0., 0.5, 0., 0.], [0.5, 0., 1., 0., 0.5, 0.], [0., 0.25, 0., 0., 0., 0.],
        [0.5, 0., 0., 0., 0.5, 1.], [0., 0.25, 0., 0.25, 0., 0.], [0., 0., 0., 0.25, 0., 0.]])
    big = mark(300)
    assert big.shape == (45150, 45150) and big.nnz == 2 * 300 * 299
    np.testing.assert_allclose(np.asarray(big.sum(axis=0)).ravel(), 1.0)  # column-stochastic


def test_laplace_generators():
    from arnoldi_amd import matrices

    g = load_golden("g1_matrices")
    np.testing.assert_array_equal(sp.csr_matrix(matrices.laplace(5)).toarray(), csr_from(g, "laplace5").toarray())
    np.testing.assert_array_equal(matrices.laplace_eigen(100), g["laplace_eigen100"])
    L1 = lambda k: sp.csr_matrix(matrices.laplace(k))  # noqa: E731
    I = sp.identity  # noqa: E741
    A2 = matrices.laplace2d(7, 9)
    want2 = sp.kron(I(9), L1(7)) + sp.kron(L1(9), I(7))
    assert (A2 - want2).nnz == 0 and A2.has_canonical_format and A2.indices.dtype == np.int32
    A3 = matrices.laplace3d(4, 5, 6)
    want3 = (sp.kron(I(30), L1(4)) + sp.kron(I(6), sp.kron(L1(5), I(4))) + sp.kron(L1(6), I(20)))
    assert (A3 - want3).nnz == 0
    g7 = load_golden("g7_laplace2d")
    assert (matrices.laplace2d(30, 31) - csr_from(g7, "lap")).nnz == 0
    part = matrices.laplace_rows((4, 5, 6), 17, 80)
    assert (part - A3[17:80]).nnz == 0


def test_shell_csr_has_the_structure_of_a_shell_fem_matrix():
    """matrices.shell_csr (the structured stand-in for BASELINE config 3): 5 unknowns per node, dense 5 x 5 blocks to
    the node itself and its six mesh neighbours -- against a brute-force assembly; symmetric pattern, canonical CSR,
    35 entries per interior row, planted diagonal."""
    from arnoldi_amd.matrices import shell_csr

    nx, ny, dof = 7, 6, 5
    A = shell_csr(nx, ny, dof, seed=3, planted=(50.0, 40.0))
    n = nx * ny * dof
    assert A.shape == (n, n) and A.has_canonical_format and A.indices.dtype == np.int32 and A.indptr.dtype == np.int32
    want = np.zeros((n, n), bool)
    for j in range(ny):
        for i in range(nx):
            for di, dj in ((0, 0), (1, 0), (-1, 0), (0, 1), (0, -1), (1, -1), (-1, 1)):
                ii, jj = i + di, j + dj
                if 0 <= ii < nx and 0 <= jj < ny:
                    a, b = dof * (j * nx + i), dof * (jj * nx + ii)
                    want[a:a + dof, b:b + dof] = True
    got = np.zeros((n, n), bool)
    got[A.nonzero()] = True
    np.testing.assert_array_equal(got, want)
    assert (want == want.T).all() and np.diff(A.indptr).max() == 7 * dof
    assert abs(A - A.T).max() > 0.1                                   # values are not symmetric
    d = A.diagonal()
    assert sorted(d)[-2:] == [40.0, 50.0] and d.min() > 0
    off = A.toarray()[~np.eye(n, dtype=bool) & want]
    assert off.min() >= -1.0 and off.max() <= 1.0
    B = shell_csr(nx, ny, dof, seed=3, planted=(50.0, 40.0))
    assert (A != B).nnz == 0                                          # same seed, same matrix


def test_random_csr_row_ranges_are_consistent():
    from arnoldi_amd.matrices import random_csr

    full = random_csr(5000, 5, 1234, planted=(4.0, 3.0))
    assert full.shape == (5000, 5000) and full.has_sorted_indices
    parts = [random_csr(5000, 5, 1234, planted=(4.0, 3.0), row_range=r) for r in ((0, 1700), (1700, 5000))]
    assert (sp.vstack(parts) - full).nnz == 0
    assert np.sort(full.diagonal())[-2:].tolist() == [3.0, 4.0]


# ---------------------------------------------------------------------------- host dense step
def test_ordered_schur_golden_and_reference_test():
    from arnoldi_amd.utils import arg_largest_magnitude, arg_largest_real, ordered_schur

    g = load_golden("g6_ordered_schur")
    for ch in ("F", "D"):  # tests/test_utils.py:22-49
        a = g[f"{ch}_a"]
        T, Z = ordered_schur(a, output="complex", sort_function=lambda v: np.argsort(v))
        tol = 3000 * np.finfo(np.float32).eps if ch == "F" else 2000 * np.finfo(np.float64).eps
        assert T.dtype == np.dtype(ch) and Z.dtype == np.dtype(ch)
        np.testing.assert_allclose(Z @ T @ Z.T.conj(), a, rtol=tol, atol=tol)
        np.testing.assert_allclose(np.diag(T), [1, 2, 3, 4, 5], rtol=tol, atol=tol)
    for tag, fn in (("lm", arg_largest_magnitude), ("lr", arg_largest_real)):
        T, Z = ordered_schur(g["hess_a"], output="complex", sort_function=fn)
        np.testing.assert_allclose(T, g[f"hess_{tag}_T"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(Z, g[f"hess_{tag}_Z"], rtol=1e-12, atol=1e-13)
    with pytest.raises(ValueError, match="needs a real matrix"):
        ordered_schur(np.eye(3) + 1j, output="real")
    with pytest.raises(ValueError, match="'complex' or 'real'"):
        ordered_schur(np.eye(3), output="quasi")


@pytest.mark.parametrize("dtype", ["f", "d"])
def test_ordered_schur_real_form(dtype):
    """The reference's own test of ``ordered_schur(output="real")`` (tests/test_utils.py:51-87), an xfail there
    ("real mode not implemented yet"), restated and passing: real orthogonal Q, quasi-triangular T of the input's
    type, ``Q T Q^T = A``, conjugate pairs as 2x2 blocks in the order of the sort function."""
    from arnoldi_amd.krylov_schur_real import real_blocks
    from arnoldi_amd.utils import ordered_schur

    r_T = np.array([[1.0, 1.5, 0.8, 0.1, 0.4],
                    [0.0, 2.0, 1.2, 1.0, 0.5],
                    [0.0, -0.3, 2.0, 1.0, 0.3],
                    [0.0, 0.0, 0.0, 4.0, 1.0],
                    [0.0, 0.0, 0.0, -2.0, 4.0]]).astype(dtype)
    complex_dtype = np.result_type(dtype, 1j)
    r_eivals = np.array([4 + 1j * np.sqrt(2), 4 - 1j * np.sqrt(2), 2 + 1j * np.sqrt(1.2 * 0.3),
                         2 - 1j * np.sqrt(1.2 * 0.3), 1]).astype(complex_dtype)
    rng = np.random.RandomState(7)
    r_Q, _ = np.linalg.qr(rng.randn(*r_T.shape).astype(dtype))
    A = r_Q.T @ r_T @ r_Q
    tol = 3000 * np.finfo(np.float32).eps if dtype == "f" else 2000 * np.finfo(np.float64).eps
    T, Q = ordered_schur(A, output="real", sort_function=lambda v: np.argsort(-np.abs(v)))
    assert T.dtype == np.dtype(dtype) and Q.dtype == np.dtype(dtype)
    np.testing.assert_allclose(Q @ T @ Q.T.conj(), A, rtol=tol, atol=tol)
    np.testing.assert_allclose(np.linalg.eigvals(T), r_eivals, rtol=tol, atol=tol)
    assert real_blocks(T) == [(0, 2), (2, 2), (4, 1)]
    np.testing.assert_allclose(Q.T @ Q, np.eye(5), atol=tol)
    # smallest first: the real eigenvalue, then the pairs by modulus
    T2, Q2 = ordered_schur(A, output="real", sort_function=lambda v: np.argsort(np.abs(v)))
    assert real_blocks(T2) == [(0, 1), (1, 2), (3, 2)]
    np.testing.assert_allclose(np.linalg.eigvals(T2), r_eivals[[4, 2, 3, 0, 1]], rtol=tol, atol=tol)
    np.testing.assert_allclose(Q2 @ T2 @ Q2.T, A, rtol=tol, atol=tol)


def test_start_vector_stream_and_history():
    from arnoldi_amd.explicit_restarts import History
    from arnoldi_amd.utils import rand_normalized_vector

    g = load_golden("g3_markov")
    np.random.seed(0)
    np.testing.assert_array_equal(rand_normalized_vector(1275, C128), g["mark50_s0_v0"])
    h = History.from_k(4)
    assert h.k == 4 and h.matvecs.dtype == np.int32 and h.restarts.dtype == np.int32
    h.matvecs[:] = [1, 2, 3, 4]
    assert h.total_matvecs == 10 and h.restarts.sum() == 0


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import arnoldi_amd
    from arnoldi_amd import _hip
    from arnoldi_amd.matrices import mark

    with pytest.raises(_hip.HipLibraryError, match="no CPU fallback"):
        arnoldi_amd.partial_schur(mark(10), 3, max_dim=5)


def test_device_init_reaches_the_library_without_a_gpu(monkeypatch):
    """_hip.device_init (first thing every device object does) must load the library and call aks_device_init --
    which, here, reports the missing GPU -- rather than block (it once took _hip's load lock twice)."""
    import torch
    from arnoldi_amd import _hip

    class _Dev:
        def __init__(self, index):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    monkeypatch.setattr(torch.cuda, "device", _Dev)
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "_devices_ready", set())
    with pytest.raises(_hip.HipLibraryError, match="aks_device_init failed"):
        _hip.device_init(0)
    assert not _hip._devices_ready


# ---------------------------------------------------------------------------- driver over the fake device
@pytest.fixture
def fake(monkeypatch):
    import fake_hip

    return fake_hip.install(monkeypatch)


def _check(A, g, prefix, seed, fake, same_restarts=True, **kw):
    import arnoldi_amd

    np.random.seed(seed)
    stats = {}
    Q, T, hist = arnoldi_amd.partial_schur(A, stats=stats, **kw)
    if same_restarts:
        np.testing.assert_array_equal(hist.restarts, g[prefix + "hist_restarts"])
        np.testing.assert_array_equal(hist.matvecs, g[prefix + "hist_matvecs"])
    np.testing.assert_allclose(np.diag(T), np.diag(g[prefix + "T"]), rtol=1e-9, atol=1e-12)
    _, _, rel = oracle.eig_residuals(A, Q, T)
    assert rel.max() <= max(1.05 * g[prefix + "rel_residuals"].max(), 1e-13)
    assert Q.flags.f_contiguous and Q.shape == (A.shape[0], kw["nev"])
    return stats


def test_driver_control_flow_against_golden(fake):
    g, g1 = load_golden("g3_markov"), load_golden("g1_matrices")
    LR = oracle.arg_largest_real
    _check(csr_from(g1, "mark10"), g, "mark10_s0_", 0, fake, nev=3, max_dim=5, sort_function=LR, max_restarts=1000)
    st = _check(csr_from(g1, "mark50"), g, "mark50_s0_", 0, fake, nev=5, max_dim=20, stopping_criterion=1e-8,
                sort_function=LR)
    assert st["matvecs"] == 20 + 20 * 10 and st["restarts"] == 21
    assert "expand" in fake.calls and "truncate" in fake.calls
    _check(csr_from(g1, "mark50"), g, "mark50_defaults_", 2, fake, nev=4, sort_function=LR)
    gd = load_golden("g2_dense_diag")
    _check(gd["diag_A"], gd, "diag_", 0, fake, same_restarts=False, nev=3, max_dim=6, sort_function=LR,
           max_restarts=1000)


def test_host_blas_runs_single_threaded_inside_a_solve(fake, monkeypatch):
    """utils.host_blas_threads: the m x m LAPACK step of a restart runs with the BLAS pools limited to one thread
    (a pool of one thread per visible CPU, woken for a 20 x 20 Schur form, doubled the time of a restart at the
    8-GPU shard sizes on some runs: profiles/r03_host_gap.txt), the pools are restored afterwards, and
    AKS_HOST_BLAS_THREADS=keep leaves them alone."""
    threadpoolctl = pytest.importorskip("threadpoolctl")
    import arnoldi_amd
    from arnoldi_amd.krylov_schur import KrylovSchurSolver

    def blas_threads():
        return [m["num_threads"] for m in threadpoolctl.threadpool_info() if m["user_api"] == "blas"]

    before = blas_threads()
    if not before or max(before) == 1:
        pytest.skip("the BLAS of this interpreter has one thread anyway")
    seen = []
    contract = KrylovSchurSolver.contract
    monkeypatch.setattr(KrylovSchurSolver, "contract", lambda self, r: (seen.append(blas_threads()), contract(self, r))[1])
    A = csr_from(load_golden("g1_matrices"), "mark50")
    for mode, expect in ((None, [1] * len(before)), ("keep", before), ("2", [min(2, b) for b in before])):
        monkeypatch.delenv("AKS_HOST_BLAS_THREADS", raising=False)
        if mode is not None:
            monkeypatch.setenv("AKS_HOST_BLAS_THREADS", mode)
        del seen[:]
        np.random.seed(0)
        arnoldi_amd.partial_schur(A, 5, max_dim=20, stopping_criterion=1e-8, sort_function=oracle.arg_largest_real)
        assert seen and all(s == expect for s in seen), (mode, seen[:2], expect)
        assert blas_threads() == before


def test_operator_formats_and_value_types(fake):
    """What the reference accepts as ``A`` (operator protocol): CSR / CSC / COO / dense, float32 /
    float64 / complex64 values.  The default tolerance follows ``A.dtype`` (krylov_schur.py:16-17: float32 =>
    3.45e-4), the work arrays are complex128 whatever the input; an integer matrix fails as in the reference."""
    import arnoldi_amd
    import scipy.sparse as sp
    from arnoldi_amd import matrices

    base = matrices.mark(30)
    shifted = (base + 0.1j * sp.diags_array(np.linspace(0, 1, base.shape[0]))).tocsr()
    LR = oracle.arg_largest_real
    for name, A in (("float32", base.astype(np.float32)), ("complex64", shifted.astype(np.complex64)),
                    ("csc", base.tocsc()), ("coo", base.tocoo()), ("dense float32", base.toarray().astype(np.float32))):
        np.random.seed(0)
        st = {}
        Q, T, h = arnoldi_amd.partial_schur(A, 4, max_dim=12, sort_function=LR, stats=st, max_restarts=300)
        np.random.seed(0)
        Qo, To, ho = oracle.krylov_schur(A, 4, max_dim=12, sort_function=LR, max_restarts=300)
        assert st["tol"] == np.sqrt(np.finfo(A.dtype).eps), name
        np.testing.assert_array_equal(h.restarts, ho.restarts, err_msg=name)
        np.testing.assert_allclose(np.diag(T), np.diag(To), rtol=1e-9, atol=1e-12, err_msg=name)
        assert Q.dtype == np.complex128 and T.dtype == np.complex128
    with pytest.raises(ValueError, match="not inexact"):
        arnoldi_amd.partial_schur(base.astype(np.int64), 4, max_dim=12)


def test_driver_errors_and_defaults(fake):
    import arnoldi_amd
    from arnoldi_amd.matrices import random_csr

    np.random.seed(0)
    with pytest.raises(ValueError, match="^Has not converged !$"):
        arnoldi_amd.partial_schur(random_csr(2000, 5, 1234), 5, max_dim=20, max_restarts=3)
    with pytest.raises(ValueError, match="^Happy breakdown not supported yet$"):
        arnoldi_amd.partial_schur(sp.identity(50, format="csr"), 2, max_dim=6)
    with pytest.raises(AssertionError):
        arnoldi_amd.partial_schur(sp.identity(50, format="csr"), 5, max_dim=4)   # nev <= p < max_dim
    with pytest.raises(AssertionError):
        arnoldi_amd.partial_schur(sp.identity(50, format="csr"), 2, max_restarts=0)
    # defaults: max_dim = min(max(2 nev + 1, 20), n), p = min(nev + 5, max_dim - 1)
    st = {}
    np.random.seed(1)
    arnoldi_amd.partial_schur(oracle.mark_matrix(20), 12, sort_function=oracle.arg_largest_real,
                              max_restarts=500, stats=st)
    assert (st["max_dim"], st["p"]) == (25, 17)
    assert st["tol"] == np.sqrt(np.finfo(np.float64).eps)


def test_chained_and_host_operator_paths(fake):
    """The Python-chained stage path (what several ranks run) and the opaque-operator path give
    the same iterates as the C-chained expansion."""
    from scipy.sparse.linalg import aslinearoperator
    from arnoldi_amd.engine import ArnoldiContext, CsrOperator, HostOperator

    A = oracle.mark_matrix(30)
    n, m = A.shape[0], 12
    np.random.seed(3)
    v0 = oracle.random_unit_vector(n, C128)
    outs = []
    for kind in ("native", "chained", "host"):
        op = HostOperator(aslinearoperator(A)) if kind == "host" else CsrOperator(A)
        ctx = ArnoldiContext(op, m)
        ctx.force_chained = kind == "chained"
        ctx.set_start_vector(v0)
        H = np.zeros((m + 1, m), C128)
        assert ctx.expand(H, 0, 5, 1e-8) == 5
        assert ctx.expand(H, 5, m, 1e-8) == m
        outs.append((H, ctx.local_columns(0, m + 1)))
    Vr = np.zeros((n, m + 1), C128, order="F")
    Hr = np.zeros((m + 1, m), C128)
    Vr[:, 0] = v0
    oracle.arnoldi_expand(A, Vr, Hr, 1e-8)
    for H, V in outs:
        np.testing.assert_allclose(H, Hr, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(V, Vr, rtol=1e-10, atol=1e-13)


def test_seam_wrappers_on_numpy_arrays(fake):
    from arnoldi_amd.decomposition import arnoldi_decomposition
    from arnoldi_amd.ortho import dgks_gs

    g = load_golden("g4_arnoldi")
    A = csr_from(g, "cplx")
    V = np.zeros((10, 7), C128)
    H = np.zeros((7, 6), C128)
    V[:, 0] = g["cplx_v0"]
    Va, Ha, k = arnoldi_decomposition(A, V, H, 1e-8)
    assert k == 6 and Va.shape == (10, 7) and Ha.shape == (7, 6)
    np.testing.assert_allclose(V, g["cplx_V"], rtol=1e-10, atol=1e-13)
    Vb, Hb = np.zeros((10, 7), C128, order="F"), np.zeros((7, 6), C128)
    Vb[:, 0] = g["brk_v0"]
    Vv, Hv, k = arnoldi_decomposition(A, Vb, Hb, 1e-8)
    assert k == 1 and Vv.shape == (10, 2) and Hv.shape == (2, 1) and Hb[1, 0] == 0

    g5 = load_golden("g5_dgks_gs")
    for tag in ("generic", "near", "inside"):
        w, h, info = g5[f"{tag}_w_in"].copy(), np.zeros(9, C128), {}
        beta, broke = dgks_gs(w, g5["V"], h, 1e-8, info=info)
        assert broke == bool(g5[f"{tag}_breakdown"])
        np.testing.assert_allclose(h, g5[f"{tag}_h"], rtol=1e-11, atol=1e-13)
        if not broke:
            np.testing.assert_allclose(w, g5[f"{tag}_w_out"], rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(beta, g5[f"{tag}_beta"], rtol=1e-10)


# ---------------------------------------------------------------------------- row sharding
def test_row_offsets():
    from arnoldi_amd.dist import row_offsets

    assert row_offsets(10, 3).tolist() == [0, 4, 7, 10]
    assert row_offsets(8, 8).tolist() == list(range(9))
    indptr = np.concatenate([[0], np.cumsum(np.r_[np.full(100, 50), np.full(900, 1)])])
    offs = row_offsets(1000, 2, indptr)
    assert offs[0] == 0 and offs[-1] == 1000 and 50 < offs[1] < 500  # heavy rows -> fewer of them
    assert np.all(np.diff(row_offsets(5, 8)) >= 0)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_split_local_rows_reassembles_the_product(world):
    from arnoldi_amd.dist import row_offsets, split_local_rows
    from arnoldi_amd.matrices import laplace2d, random_csr

    rng = np.random.default_rng(world)
    for A in (random_csr(900, 5, 7), laplace2d(20, 23), sp.csr_matrix(oracle.mark_matrix(25))):
        n = A.shape[0]
        offs = row_offsets(n, world, A.indptr)
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        y = np.zeros(n, C128)
        for r in range(world):
            r0, r1 = offs[r], offs[r + 1]
            plan = split_local_rows(A[r0:r1], offs, r)
            assert plan.diag.shape == (r1 - r0, max(r1 - r0, 1)) and plan.recv_counts.sum() == plan.n_ghost
            assert np.all((plan.ghost_cols < r0) | (plan.ghost_cols >= r1))
            owners = np.searchsorted(offs, plan.ghost_cols, side="right") - 1
            assert np.all(np.diff(owners) >= 0) and plan.recv_counts[r] == 0
            part = plan.diag @ x[r0:r1]
            if plan.off is not None:
                part = part + plan.off @ x[plan.ghost_cols]
            y[r0:r1] = part
        np.testing.assert_allclose(y, A @ x, rtol=1e-13, atol=1e-13)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_dist_worker(tmp_path, nproc, backend, device, timeout=600, extra=()):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), "--backend", backend, "--device", device,
           "--out", str(tmp_path)] + list(extra)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return [json.load(open(os.path.join(tmp_path, f"rank{r}.json"))) for r in range(nproc)]


def run_hostcomm_worker(tmp_path, nproc, case, timeout=600):
    """``nproc`` plain processes (no launcher, no torch): tests/hostcomm_worker.py over ``dist.HostComm``."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(OMP_NUM_THREADS="2", WORLD_SIZE=str(nproc), AKS_RENDEZVOUS=f"127.0.0.1:{_free_port()}", AKS_COMM_TIMEOUT_S="120")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "hostcomm_worker.py"), "--case", case, "--out", str(tmp_path)],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(nproc)]
    outs = []
    for p_ in procs:
        try:
            outs.append(p_.communicate(timeout=timeout)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p_.returncode == 0 for p_ in procs), "\n".join(o[-3000:] for o in outs)
    return [json.load(open(os.path.join(tmp_path, f"rank{r}.json"))) for r in range(nproc)]


@pytest.mark.parametrize("ranks", [2, 5])
def test_torch_free_ranks_set_up_over_the_tcp_rendezvous(tmp_path, ranks):
    """dist.HostComm: process ranks without torch.distributed -- and without torch in the process at
    all -- carry the set-up exchanges of the row-sharded solve (all-gather, ghost requests, row gather, max, barrier,
    and the chained path's all-reduce / all-to-all) over a TCP rendezvous."""
    verdicts = run_hostcomm_worker(tmp_path, ranks, "setup")
    assert all(v == {"setup": "ok", "size": ranks, "torch_imported": False} for v in verdicts), verdicts


def test_rendezvous_fails_loudly_when_a_rank_is_missing(tmp_path):
    """A rank that never arrives is a RuntimeError after the time-out on the ranks that did, not a hang."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", AKS_RENDEZVOUS=f"127.0.0.1:{_free_port()}", AKS_COMM_TIMEOUT_S="2")
    code = ("import sys; sys.path.insert(0, %r); from arnoldi_amd.dist import HostComm\n"
            "try:\n    HostComm()\nexcept RuntimeError as e:\n    print('RuntimeError:', e); sys.exit(7)" % os.path.join(ROOT, "arnoldi-py_amd"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 7 and "only 1 of 2 ranks arrived" in r.stdout, r.stdout + r.stderr


def check_dist_verdicts(verdicts, native=False):
    for v in verdicts:
        assert all(c["native_comm"] == native for c in v.values() if "native_comm" in c), "unexpected collective path"
        for c in v.values():
            c.pop("native_comm", None)
        mm = v.pop("mismatch")
        assert mm["size"].startswith("ValueError: row shards disagree"), mm
        assert mm["offsets"].startswith("ValueError: row shards disagree"), mm
        rl = v.pop("real")
        for name, c in rl.items():
            assert c["eig_err"] < 10 * c["tol"] * 4.5 and c["orth_err"] < 1e-11, (name, c)
            assert c["rel"] <= max(1.05 * c["rel_oracle"], 10 * c["tol"]), (name, c)
            assert abs(c["restarts"] - c["restarts_oracle"]) <= max(3, 0.3 * c["restarts_oracle"]), (name, c)
        ex = v.pop("explicit")
        for name in ("mark30", "planted"):
            c = ex[name]
            assert c["hist_equal"] and c["eig_err"] < 1e-9, (name, c)
            assert c["res_max"] <= max(2 * c["res_oracle_max"], 1e-11), (name, c)
            assert c["device_residual_err"] < 1e-11, (name, c)
        assert ex["mark30"]["shape"] == [465, 4] and ex["planted"]["shape"] == [6000, 3]
        c = ex["naive"]
        assert c["flags_equal"] and c["value_err"] < 1e-10 and c["vector_shape"] == [55, 1], c
        assert abs(c["true_residual"] - c["true_residual_oracle"]) <= 1e-3 * c["true_residual_oracle"] + 1e-12, c
        assert set(v) == {"mark50", "laplace2d", "random_planted", "local_rows", "block_diag", "complex"}
        for name, c in v.items():
            assert c["restarts_equal"] and c["matvec_hist_equal"], (name, c)
            assert c["eig_err"] < 1e-9, (name, c)
            assert c["rel_residual"] <= max(1.05 * c["rel_residual_oracle"], 1e-13), (name, c)
            assert c["orth_err"] < 1e-12, (name, c)
            assert c["device_residual_err"] < 1e-11, (name, c)
            # third all-reduce only once a step has needed the second DGKS pass (then the expansion is redone)
            assert c["lazy_redos"] == (1 if c["second_passes"] else 0), (name, c)
            exch = 0 if name == "block_diag" else 1
            assert c["collectives_per_step"] == exch + (3 if c["second_passes"] else 2), (name, c)
        assert v["laplace2d"]["lazy_redos"] == 1 and v["block_diag"]["collectives_per_step"] in (2, 3)
        assert v["random_planted"]["n_ghost"] > 1000 and v["block_diag"]["n_ghost"] == 0
        assert 0 < v["laplace2d"]["n_ghost"] <= 60
    assert all(v == verdicts[0] or v["mark50"]["restarts"] == verdicts[0]["mark50"]["restarts"] for v in verdicts)


def test_one_comm_per_process_group():
    """dist.comm_for: repeated solves on one process group share one Comm (hence one RCCL communicator); a group that
    was destroyed and initialised again gets a new one (a new communicator per solve, never destroyed)."""
    import torch.distributed as dist
    from arnoldi_amd.dist import comm_for

    def init():
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)

    init()
    try:
        a = comm_for()
        assert comm_for() is a and a.size == 1 and a.native() is None       # gloo: no RCCL handle, chained path
    finally:
        dist.destroy_process_group()
    init()
    try:
        b = comm_for()
        assert b is not a and comm_for() is b
    finally:
        dist.destroy_process_group()


def test_row_sharded_solve_four_ranks_gloo(tmp_path):
    """world_size = 4: interior ranks exchange with two neighbours, empty messages to the others."""
    check_dist_verdicts(run_dist_worker(tmp_path, 4, "gloo", "cpu"))


def test_row_sharded_solve_two_ranks_gloo(tmp_path):
    """world_size = 2 over gloo on CPU tensors: partition, ghost exchange, all-reduces and the
    stage chaining of the multi-GPU driver, against single-process oracle solves."""
    check_dist_verdicts(run_dist_worker(tmp_path, 2, "gloo", "cpu"))


def test_host_planners_under_address_and_ub_sanitizers():
    """tests/asan: the library source with -fsanitize=address,undefined on its host code (CPU build; the pool has no
    GPU ASan), driven through every host-side planner entry point on the shapes of the SpMV form tests plus
    degenerate ones; every planned array is compared with the matrix it came from."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "asan")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    exe = os.path.join(ROOT, "tests", "asan", "planner_asan")
    syms = subprocess.run(["nm", exe], capture_output=True, text=True).stdout
    assert "__asan_report_load" in syms and "__ubsan_handle" in syms          # the instrumentation is really in
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "all planner checks passed" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_native_legacy_randn_is_numpys_stream_bit_for_bit(monkeypatch):
    """utils.legacy_randn / aks_legacy_randn (csrc/aks_host_rng.cpp, include/arnoldi_hostrng.h): the reference's start
    vector is ONE np.random.randn(n) on the global legacy generator (src/arnoldi/utils.py:10); the native helper must
    return the same bits AND leave the generator where NumPy would -- for any position in the Mersenne Twister's block,
    with and without a cached gaussian, odd and even n, around its block sizes, for 1 and several threads."""
    import ctypes as C

    from arnoldi_amd import _hip, utils

    fn = C.CDLL(_hip.LIB_PATH).aks_legacy_randn
    fn.restype = C.c_int

    def native(n):
        st = np.random.get_state()
        key = np.array(st[1], dtype=np.uint32)
        pos, hg, g = C.c_int32(st[2]), C.c_int32(st[3]), C.c_double(st[4])
        out = np.empty(n)
        assert fn(C.c_void_p(key.ctypes.data), C.byref(pos), C.byref(hg), C.byref(g), C.c_void_p(out.ctypes.data), C.c_int64(n)) == 0
        np.random.set_state(("MT19937", key, pos.value, hg.value, g.value))
        return out

    for threads in ("1", "5"):
        monkeypatch.setenv("AKS_PLAN_THREADS", threads)
        for seed, skip in ((0, 0), (7, 3), (123, 623)):
            for n in (0, 1, 2, 3, 8191, 8192, 8193, 8194, 50_001, 300_000):
                np.random.seed(seed)
                np.random.rand(skip)                         # somewhere inside the generator's block of 624 words
                if n % 3 == 1:
                    np.random.randn(1)                       # leaves a cached second value
                st = np.random.get_state()
                want, after = np.random.randn(n), np.random.randn(3)
                np.random.set_state(st)
                got, after2 = native(n), np.random.randn(3)
                assert np.array_equal(want, got) and np.array_equal(after, after2), (threads, seed, n)
    # the product's entry: native from 1M entries on, NumPy below; the start vector of the reference either way
    np.random.seed(3)
    want = np.random.randn(1_200_001)
    tail = np.random.rand(2)
    np.random.seed(3)
    got = utils.legacy_randn(1_200_001)
    assert utils._native_randn not in (None, False)          # the in-tree library exports the helper
    assert np.array_equal(want, got) and np.array_equal(tail, np.random.rand(2))
    for seed, n in ((0, 1_000_000), (5, 1_000_003)):         # the reference's two statements, verbatim, against the fast spelling
        np.random.seed(seed)
        v = utils.rand_normalized_vector(n, C128)
        np.random.seed(seed)
        w = np.random.randn(n).astype(C128)
        w /= np.linalg.norm(w)
        assert np.array_equal(v.view(np.uint64), w.view(np.uint64))      # bits, signs of zeros included
        np.random.seed(seed)
        r = utils.rand_normalized_vector(n)
        np.random.seed(seed)
        wr = np.random.randn(n).astype(np.float64)
        wr /= np.linalg.norm(wr)
        assert np.array_equal(r, wr)
    # two threads drawing at once (two solver constructors): the draws are serialised, so together they are
    # the two consecutive draws of the stream -- in either order -- never the same vector twice
    import threading

    np.random.seed(11)
    first, second = np.random.randn(1_000_001), np.random.randn(1_000_001)
    np.random.seed(11)
    out = [None, None]
    ts = [threading.Thread(target=lambda k=k: out.__setitem__(k, utils.legacy_randn(1_000_001))) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert (np.array_equal(out[0], first) and np.array_equal(out[1], second)) or \
           (np.array_equal(out[1], first) and np.array_equal(out[0], second))
    monkeypatch.setenv("AKS_NATIVE_RANDN", "0")               # switch: NumPy's own loop
    np.random.seed(3)
    assert np.array_equal(utils.legacy_randn(1_200_001), want)


def test_host_planners_are_thread_count_independent_and_race_free():
    """The binned and sliced planners split their passes over host threads (``AKS_PLAN_THREADS``, default
    min(hardware threads, 16): 0.64 -> 0.10 s for the 10M-row headline matrix, most of a call's time to solution).
    (1) The plan is the same bytes for 1, 3 and 7 threads; (2) the planner driver of tests/asan under ThreadSanitizer
    with 8 threads reports nothing."""
    import hashlib

    from arnoldi_amd import _hip
    from arnoldi_amd.matrices import mark, random_csr

    lib = _hip.load()

    def binned(M):
        indptr, indices, values = (np.ascontiguousarray(M.indptr, np.int32), np.ascontiguousarray(M.indices, np.int32),
                                   np.ascontiguousarray(M.data))
        sz = _hip.PbSizes()
        plan = lib.aks_pb_plan_create(indptr.ctypes.data, indices.ctypes.data, values.ctypes.data, int(np.iscomplexobj(values)),
                                      M.shape[0], M.shape[1], C.byref(sz))
        assert plan, lib.aks_last_error()
        out = [np.empty(sz.nnz_pad, values.dtype), np.empty(sz.nnz_pad, np.uint16), np.empty(sz.n_slabs, np.int32),
               np.empty(sz.n_slabs, np.int32), np.empty((sz.n_runs, 4), np.uint32), np.empty(sz.n_rowblocks + 1, np.int32),
               np.empty(sz.n_lrow, np.uint16)]
        assert lib.aks_pb_plan_export(plan, *(a.ctypes.data for a in out)) == 0
        lib.aks_pb_plan_destroy(plan)
        return hashlib.sha256(b"".join(a.tobytes() for a in out)).hexdigest()

    def sliced(M):
        indptr, indices, values = (np.ascontiguousarray(M.indptr, np.int32), np.ascontiguousarray(M.indices, np.int32),
                                   np.ascontiguousarray(M.data))
        n = M.shape[0]
        total = lib.aks_sell_plan_size(indptr.ctypes.data, n)
        out = [np.empty((n + 63) // 64 + 1, np.int64), np.empty(total, np.int32), np.empty(total, values.dtype)]
        assert lib.aks_sell_plan_fill(indptr.ctypes.data, indices.ctypes.data, values.ctypes.data, int(np.iscomplexobj(values)), n,
                                      *(a.ctypes.data for a in out)) == 0
        return hashlib.sha256(b"".join(a.tobytes() for a in out)).hexdigest()

    A = random_csr(300_000, 5, 7)                       # 1.5M entries: above the planners' single-thread threshold
    Ac = A.astype(C128) * (1 + 0.5j)
    Mk = mark(800)                                      # 1.28M entries, rows of 2 .. 4
    old = os.environ.get("AKS_PLAN_THREADS")
    try:
        seen = {}
        for nt in ("1", "3", "7"):
            os.environ["AKS_PLAN_THREADS"] = nt
            seen[nt] = (binned(A), binned(Ac), sliced(Mk), sliced(Ac))
        assert seen["1"] == seen["3"] == seen["7"]
    finally:
        if old is None:
            os.environ.pop("AKS_PLAN_THREADS", None)
        else:
            os.environ["AKS_PLAN_THREADS"] = old
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "asan"), "tsan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([os.path.join(ROOT, "tests", "asan", "planner_tsan")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, AKS_PLAN_THREADS="8"))
    assert r.returncode == 0 and "all planner checks passed" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "ThreadSanitizer" not in r.stderr and "ThreadSanitizer" not in r.stdout, r.stderr[-3000:]


# ---------------------------------------------------------------------------- tile-binned SpMV form
@pytest.mark.parametrize("shape,complex_vals", [((5000, 5000), False), ((3000, 200_000), True), ((70_000, 900), False),
                                                ((20_000, 20_000), False),
                                                ((2_130_000, 30_000), False)])     # 261 row blocks: > AKS_PB_CHUNKS
def test_binned_plan_represents_the_matrix(fake, shape, complex_vals):
    """aks_pb_plan_create / aks_pb_plan_export are host code: replaying the two phases on the planned
    arrays (tests/fake_hip.py) must reproduce A @ x, including empty rows, a long row, several
    sub-slabs, several row blocks and non-square shapes (the off-diagonal blocks of a row shard).
    The schedule invariants that make phase 2 reproducible are checked on the arrays themselves."""
    import torch
    from arnoldi_amd import _hip
    from arnoldi_amd.device import DeviceCSR

    rng = np.random.default_rng(shape[0])
    n_rows, n_cols = shape
    nnz = 6 * n_rows if n_rows < 1_000_000 else n_rows
    rows = rng.integers(0, n_rows, nnz)
    rows[rows % 7 == 0] = 1                       # empty rows + one long row (many runs, every level)
    cols = rng.integers(0, n_cols, nnz)
    vals = rng.standard_normal(nnz) + (1j * rng.standard_normal(nnz) if complex_vals else 0)
    A = sp.csr_matrix((vals, (rows, cols)), shape=shape)
    A.sum_duplicates()
    dA = DeviceCSR(A)
    b = dA.build_binned()
    d = b.desc
    assert d.n_slabs == -(-n_cols // 8192) and d.n_rowblocks == -(-n_rows // 8192) and d.nnz == A.nnz
    runs = b.runs.numpy().view(np.uint32).astype(np.int64)
    info = runs[:, 3]
    l0, l01, total, levels = info & 127, (info >> 7) & 127, (info >> 14) & 127, (info >> 21) & 15
    assert int(total.sum()) == A.nnz and total.max() <= 64 and levels.max() <= 8
    assert np.all(l0 <= l01) and np.all(l01 <= total)
    RPR = _hip.PB_RUNS_PER_ROUND
    assert len(runs) % RPR == 0 and np.all(total[-RPR:] == 0)          # whole rounds, the last one empty
    assert b.lanes_per_load > 32 or A.nnz < 64 * len(runs) // 4        # wave-loads are filled from several tiles
    # a round = 32 consecutive wave-loads, 4 per wave.  Walking a round in (wave, load, lane) order, the level of
    # an entry is that of the previous entry of its row if that one sits in the SAME wave (a wave's LDS
    # operations complete in order), one more otherwise (a barrier separates the two adds)
    lrow = b.lrow.numpy().view(np.uint16).astype(np.int64)
    assert len(lrow) == 64 * len(runs)
    for r0 in range(0, len(runs), RPR):
        assert len(set(levels[r0:r0 + RPR].tolist())) == 1
        last = {}
        for j in range(RPR):
            w, k = divmod(j, RPR // 8)
            for lane in range(int(total[r0 + j])):
                word = int(lrow[(r0 // RPR) * 2048 + w * 256 + lane * 4 + k])
                row, lv = word & 8191, word >> 13
                assert lv < levels[r0]
                if row in last:
                    pw, pj, plv = last[row]
                    assert lv == (plv if pw == w else plv + 1)
                else:
                    assert lv == 0
                last[row] = (w, j, lv)
        if r0 > 40 * RPR:
            break
    x = torch.from_numpy((rng.standard_normal(n_cols) + 1j * rng.standard_normal(n_cols)).astype(C128))
    y = torch.from_numpy((rng.standard_normal(n_rows) + 0j).astype(C128))
    y0 = y.numpy().copy()
    dA.use_binned = True
    dA.spmv(x, y)
    np.testing.assert_allclose(y.numpy(), A @ x.numpy(), rtol=1e-12, atol=1e-12)
    dA.spmv(x, y, accumulate=True)
    np.testing.assert_allclose(y.numpy(), 2 * (A @ x.numpy()), rtol=1e-12, atol=1e-12)
    assert "pb_spmv" in fake.calls and not np.array_equal(y0, y.numpy())


@pytest.mark.parametrize("shape,complex_vals", [((1000, 1000), False), ((130, 4000), True), ((4097, 300), False)])
def test_sliced_plan_represents_the_matrix(fake, shape, complex_vals):
    """``aks_sell_plan_size`` / ``aks_sell_plan_fill`` (host code of the library) and the NumPy replay of the sliced
    form: slices of 64 rows stored entry-major, padded with column -1; ragged last slice, empty rows, a long row."""
    import torch
    from arnoldi_amd.device import DeviceCSR

    rng = np.random.default_rng(shape[1])
    n_rows, n_cols = shape
    nnz = 5 * n_rows
    rows = rng.integers(0, n_rows, nnz)
    rows[rows % 11 == 0] = 3                      # empty rows + one long row
    cols = rng.integers(0, n_cols, nnz)
    vals = rng.standard_normal(nnz) + (1j * rng.standard_normal(nnz) if complex_vals else 0)
    A = sp.csr_matrix((vals, (rows, cols)), shape=shape)
    A.sum_duplicates()
    dA = DeviceCSR(A)
    lens = np.diff(A.indptr)
    want = sum(64 * int(lens[r0:r0 + 64].max()) for r0 in range(0, n_rows, 64))
    assert abs(dA.sliced_padding() * A.nnz - want) < 0.5
    b = dA.build_sliced()
    d = b.desc
    assert d.nnz_pad == want and d.n_slices == -(-n_rows // 64) and d.nnz == A.nnz
    slice_ptr, col, val = b.slice_ptr.numpy(), b.col.numpy(), b.val.numpy()
    for r in (0, 3, 63, 64, n_rows - 1):           # slot slice_ptr[s] + k * 64 + lane holds entry k of the row
        s_, lane = divmod(r, 64)
        k = np.arange(lens[r])
        assert np.array_equal(col[slice_ptr[s_] + k * 64 + lane], A.indices[A.indptr[r]:A.indptr[r + 1]])
        assert np.array_equal(val[slice_ptr[s_] + k * 64 + lane], A.data[A.indptr[r]:A.indptr[r + 1]])
        width = (slice_ptr[s_ + 1] - slice_ptr[s_]) // 64
        assert np.all(col[slice_ptr[s_] + np.arange(lens[r], width) * 64 + lane] == -1)
    x = torch.from_numpy((rng.standard_normal(n_cols) + 1j * rng.standard_normal(n_cols)).astype(C128))
    y = torch.from_numpy((rng.standard_normal(n_rows) + 0j).astype(C128))
    dA.form = "sliced"
    dA.spmv(x, y)
    np.testing.assert_allclose(y.numpy(), A @ x.numpy(), rtol=1e-12, atol=1e-12)
    dA.spmv(x, y, accumulate=True)
    np.testing.assert_allclose(y.numpy(), 2 * (A @ x.numpy()), rtol=1e-12, atol=1e-12)
    assert fake.calls.count("sell_spmv") == 2
    blk = dA.block()
    assert bool(blk.sell) and not bool(blk.pb)
    if not complex_vals:
        xr, yr = torch.from_numpy(rng.standard_normal(n_cols)), torch.zeros(n_rows, dtype=torch.float64)
        dA.spmv(xr, yr, real=True)
        np.testing.assert_allclose(yr.numpy(), A @ xr.numpy(), rtol=1e-12, atol=1e-12)


def test_solver_runs_on_the_sliced_form(fake):
    """``spmv_form="sliced"``: the operator hands aks_arnoldi_expand a block whose ``sell`` pointer is set, and the
    solve matches the CSR-stream solve."""
    import arnoldi_amd
    from arnoldi_amd.engine import CsrOperator
    from arnoldi_amd.matrices import mark
    from arnoldi_amd.utils import arg_largest_real

    A = mark(12)
    ref = arnoldi_amd.partial_schur(A, 3, max_dim=12, sort_function=arg_largest_real, stopping_criterion=1e-9)
    op = CsrOperator(A, spmv_form="sliced")
    assert op.spmv_form == "sliced" and bool(op.shard.diag.sell)
    Q, T, hist = arnoldi_amd.partial_schur(op, 3, max_dim=12, sort_function=arg_largest_real, stopping_criterion=1e-9)
    np.testing.assert_allclose(np.sort_complex(np.diag(T)), np.sort_complex(np.diag(ref[1])), rtol=1e-9, atol=1e-12)
    assert any(c.startswith("sell_spmv") for c in fake.calls)


def test_binned_plan_refuses_what_it_cannot_index():
    """Too many (sub-slab, row block) tiles: the planner reports it through a NULL plan + aks_last_error and
    ``DeviceCSR.autotune`` keeps the CSR-stream kernel."""
    from arnoldi_amd import _hip

    lib = _hip.load()
    n = 8192 * 9000                                # 9000 x 9000 tiles > 2^26
    indptr = np.zeros(2, np.int32)                 # (the shape check comes before any array is read)
    sz = _hip.PbSizes()
    plan = lib.aks_pb_plan_create(indptr.ctypes.data, indptr.ctypes.data, indptr.ctypes.data, 0, n, n, C.byref(sz))
    assert not plan and b"tiles" in lib.aks_last_error()


def test_scatter_ratio_separates_stencils_from_random_graphs():
    from arnoldi_amd import matrices
    from arnoldi_amd.device import DeviceCSR

    ratio = lambda A: DeviceCSR.scatter_ratio(type("S", (), {"_host": sp.csr_matrix(A), "nnz": A.nnz,  # noqa: E731
                                                              "n_rows": A.shape[0]})())
    assert ratio(matrices.laplace2d(300, 301)) < 0.2
    assert ratio(matrices.laplace3d(40, 41, 42)) < 0.3
    assert ratio(matrices.random_csr(200_000, 5, 1)) > 0.9


# ---------------------------------------------------------------- explicit restarts (host logic)
def test_explicit_restart_building_blocks(fake):
    import explicit_cases as ec

    ec.check_ritz_decomposition()
    ec.check_ritz_wide(n=500, m=100, q=100)
    ec.check_mgs()
    ec.check_ritz_reference_tests()
    assert "combine" in fake.calls


def test_naive_explicit_restarts_host_logic(fake):
    import explicit_cases as ec

    ec.check_naive()


@pytest.mark.parametrize("tag", ["defl_mark10", "defl_diag", "defl_mark30", "defl_lap"])
def test_explicit_restarts_with_deflation_host_logic(fake, tag):
    import explicit_cases as ec

    ec.check_deflation(tag)


def test_explicit_restarts_reference_tests_host_logic(fake):
    import explicit_cases as ec

    ec.check_deflation_reference_tests()


def test_lookahead_application_is_used_and_changes_nothing(fake, monkeypatch):
    """The Krylov-Schur driver queues A V[:, m] behind the copy of H (engine.ArnoldiContext.expand) and
    starts the next expansion from it (AKS_EXPAND_FROM_W): same iterates as without."""
    import arnoldi_amd
    from arnoldi_amd import matrices
    from arnoldi_amd.utils import arg_largest_real

    A = matrices.mark(20)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("AKS_LOOKAHEAD", flag)
        fake.calls.clear()
        np.random.seed(3)
        st = {}
        Q, T, h = arnoldi_amd.partial_schur(A, 3, max_dim=12, stopping_criterion=1e-9, sort_function=arg_largest_real,
                                            stats=st)
        out[flag] = (Q, T, h.restarts.copy(), st["matvecs"], st["lookahead_applies"], list(fake.calls))
    on, off = out["1"], out["0"]
    assert "expand_from_w" in on[5] and "expand_from_w" not in off[5]
    assert on[4] == int(on[2].max()) and off[4] == 0     # one per expansion; none when switched off
    np.testing.assert_array_equal(on[0], off[0])
    np.testing.assert_array_equal(on[1], off[1])
    np.testing.assert_array_equal(on[2], off[2])
    assert on[3] == off[3]                      # Arnoldi steps consumed, not speculative applications


def test_happy_breakdown_deflate_host_logic(fake):
    import explicit_cases as ec

    ec.check_happy_breakdown_deflate()


# ---------------------------------------------------------------- real-arithmetic mode (host logic)
def test_reorder_real_schur():
    import real_cases as rc

    rc.check_reorder_real_schur()


@pytest.mark.parametrize("name", ["mark30_lr", "mark50_readme", "planted_odd_n", "laplace2d", "conjugate_pairs",
                                  "pair_cut_at_nev3", "pair_cut_at_nev5", "dense_array"])
def test_real_arithmetic_host_logic(fake, name):
    import real_cases as rc

    A, nev, seed, kw = rc.cases()[name]
    rc.check_case(A, nev, seed, **kw)
    assert any(c.endswith("_real") for c in fake.calls)


def test_real_arithmetic_errors(fake):
    import real_cases as rc

    rc.check_errors()
    rc.check_auto()


@pytest.mark.parametrize("name", ["mark50_readme", "laplace2d", "pair_cut_at_nev5", "conjugate_pairs"])
def test_real_arithmetic_locking_host_logic(fake, name):
    import real_cases as rc

    A, nev, seed, kw = rc.cases()[name]
    rc.check_locking_real(A, nev, seed, **{k: v for k, v in kw.items() if k != "max_restarts"})


def test_real_arithmetic_deflate_and_residuals_host_logic(fake):
    import real_cases as rc

    rc.check_deflate_real()
    rc.check_residual_norms_real()


@pytest.mark.parametrize("name", ["mark30_lr", "pair_cut_at_nev3", "planted_odd_n"])
def test_real_arithmetic_explicit_restarts_host_logic(fake, name):
    import real_cases as rc

    A, nev, seed, kw = rc.cases()[name]
    kw = {k: v for k, v in kw.items() if k != "max_restarts"}
    rc.check_explicit_deflation_real(A, min(nev, 3), seed, max_restarts=400, **kw)


# ---------------------------------------------------------------------------- locking + dynamic p
@pytest.mark.parametrize("case", ["mark50_lr", "laplace_lm", "planted_lm"])
def test_locking_gives_the_oracles_eigenpairs(fake, case):
    """``partial_schur(..., locking=True)`` (the reference's TODO, README.md:116) against the oracle of the
    reference's algorithm: same wanted eigenvalues, partial Schur relation and residual bound; locked vectors
    leave the restart compression (fewer panel bytes per restart) and p grows with them."""
    import arnoldi_amd
    from arnoldi_amd import matrices

    if case == "mark50_lr":
        A, nev, kw, sort_o = matrices.mark(50), 5, dict(max_dim=20, stopping_criterion=1e-8), oracle.arg_largest_real
    elif case == "laplace_lm":
        A, nev, kw, sort_o = matrices.laplace2d(30, 31), 6, dict(max_dim=30), oracle.arg_largest_magnitude
    else:
        A = matrices.random_csr(6000, 5, 1234, planted=(4.0, 3.7, 3.4, 3.1, 2.8, 2.5))
        nev, kw, sort_o = 5, dict(max_dim=20), oracle.arg_largest_magnitude
    np.random.seed(0)
    Qo, To, ho = oracle.krylov_schur(A, nev, sort_function=sort_o, max_restarts=2000, **kw)
    np.random.seed(0)
    st = {}
    Q, T, h = arnoldi_amd.partial_schur(A, nev, sort_function=sort_o, locking=True, stats=st, max_restarts=2000, **kw)
    tol = st["tol"]
    assert st["locked"] == nev and Q.shape == (A.shape[0], nev)
    np.testing.assert_allclose(np.sort_complex(np.diag(T)), np.sort_complex(np.diag(To)), rtol=50 * tol, atol=50 * tol)
    assert np.abs(np.tril(T, -1)).max() == 0
    np.testing.assert_allclose(Q.conj().T @ Q, np.eye(nev), atol=1e-11)
    # A Q = Q T up to the deflated couplings (each < tol |theta|)
    scale = np.abs(np.diag(T)).max()
    assert np.linalg.norm(A @ Q - Q @ T, axis=0).max() < 5 * tol * scale * np.sqrt(nev)
    _, _, rel = oracle.eig_residuals(A, Q, T)
    _, _, rel_o = oracle.eig_residuals(A, Qo, To)
    assert rel.max() <= max(1.05 * rel_o.max(), 10 * tol), (rel.max(), rel_o.max())
    # restart count in the oracle's ballpark, every History entry set once
    assert 0 < st["restarts"] <= 2 * int(ho.restarts.max()) + 5
    assert np.all(h.restarts > 0) and np.all(np.diff(h.restarts) >= 0)
    # the compression moved fewer bytes once values were locked: 16 n (m + p - 2 l + 2) with growing l
    tb = st["truncation_bytes"]
    n, m, p0 = A.shape[0], st["max_dim"], min(nev + 5, st["max_dim"] - 1)
    assert tb[0] == 16 * n * (m + p0 + 2) and min(tb) < tb[0]
    assert st["p"] >= p0


# ---------------------------------------------------------------------------- lint of the compiled ISA
def test_scalar_load_hazard_lint():
    """csrc/check_scalar_hazards.py (``make -C arnoldi-py_amd hazards``): (1) it flags the ISA round 3's
    ``k_colscale_after_truncate`` compiled to -- a scalar load of ``cs[m]`` waited for only behind the vector stores that
    clear ``cs[m]``: the lost carried scale of deferred normalisation, found at full size in round 4; (2) synthetic
    listings for each rule: fields of one struct pass, overlapping bytes / a rewritten base / a lane offset / an address
    that cannot be followed do not; a scalar load that FOLLOWS a store of the same argument is found through a loop's back
    edge and through s_mul- and s_lshl-formed offsets; (3) the kernels of the current tree compile
    to ISA without any candidate, with EVERY scalar data load and vector store followed to a kernel argument."""
    import re
    import subprocess
    import sys

    from conftest import ROOT

    tool = os.path.join(ROOT, "arnoldi-py_amd", "csrc", "check_scalar_hazards.py")
    bad = subprocess.run([sys.executable, tool, os.path.join(ROOT, "tests", "golden", "k_colscale_r03.s")], capture_output=True, text=True)
    assert bad.returncode == 1 and "k_colscale_after_truncate" in bad.stdout and "can overtake it" in bad.stdout, bad.stdout + bad.stderr
    import importlib.util

    spec = importlib.util.spec_from_file_location("check_scalar_hazards", tool)
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    # argument 0 is a pointer P (s[16:17]); the scalar data load reads bytes 8..15 of *P
    head = ["s_load_dwordx2 s[16:17], s[0:1], 0x0", "s_waitcnt lgkmcnt(0)", "s_load_dwordx2 s[12:13], s[16:17], 0x8", "v_mov_b32_e32 v6, 0"]

    def verdict(*tail, head=head):
        body = list(enumerate(list(head) + list(tail) + ["s_waitcnt lgkmcnt(0)", "s_endpgm"], 1))
        return [f[2] for f in lint.check_kernel("k", body)]

    assert verdict("global_store_dwordx4 v6, v[2:5], s[16:17] offset:16", "global_store_dword v6, v7, s[16:17] offset:36") == []
    assert len(verdict("global_store_dwordx4 v6, v[2:5], s[16:17] offset:4")) == 1                 # bytes 4..19 overlap 8..15
    assert len(verdict("global_store_dword v6, v7, s[16:17] offset:8")) == 1
    assert len(verdict("s_add_u32 s16, s16, 8", "global_store_dword v6, v7, s[16:17] offset:36")) == 1   # base rewritten
    assert len(verdict("v_mov_b32_e32 v6, v9", "global_store_dword v6, v7, s[16:17] offset:36")) == 1    # lane offset not 0
    unfollowed = verdict("global_store_dword v[8:9], v7, off")                                     # an address from nowhere:
    assert any("cannot be told apart" in f for f in unfollowed) and any("could be followed" in f for f in unfollowed)
    assert verdict("s_waitcnt lgkmcnt(0)", "global_store_dword v6, v7, s[16:17] offset:8") == []   # waited for first
    # a store through another argument is unrelated, whatever the order
    assert verdict("s_load_dwordx2 s[18:19], s[0:1], 0x8", "s_waitcnt lgkmcnt(0)", "global_store_dword v6, v7, s[18:19] offset:8",
                   "s_load_dword s30, s[16:17], 0x8") == []
    # rule (A) through a loop: iteration i stores to P + 24 i + 8, iteration i + 1 scalar-loads P + 24 (i + 1) -- the offset
    # is s_mul-formed and the base is redefined inside the loop, so nothing proves the bytes disjoint
    loop = ["s_load_dwordx2 s[16:17], s[0:1], 0x0", "s_load_dword s20, s[0:1], 0x8", "s_waitcnt lgkmcnt(0)", "s_mov_b32 s21, 0",
            ".LBB0_1:", "s_mul_i32 s22, s21, 24", "s_mul_hi_u32 s23, s21, 24", "s_add_u32 s24, s16, s22", "s_addc_u32 s25, s17, s23",
            "s_load_dwordx2 s[12:13], s[24:25], 0x0", "s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v2, s12", "v_mov_b32_e32 v3, s13",
            "v_mov_b32_e32 v6, 0", "global_store_dwordx2 v6, v[2:3], s[24:25] offset:8", "s_add_i32 s21, s21, 1",
            "s_cmp_lt_i32 s21, s20", "s_cbranch_scc1 .LBB0_1"]
    found = verdict(head=loop)
    assert len(found) == 1 and "can execute AFTER the vector store" in found[0], found
    straight = [x for x in loop if not x.startswith((".LBB", "s_cbranch"))]                         # once, no back edge: the load is
    assert verdict(head=straight) == []                                                            # complete before the store, never after
    # rule (A) behind an s_lshl-formed offset: store to P + 8 n, then a scalar load of P[0]
    after = ["s_load_dwordx2 s[16:17], s[0:1], 0x0", "s_load_dwordx2 s[20:21], s[0:1], 0x8", "s_waitcnt lgkmcnt(0)",
             "s_lshl_b64 s[22:23], s[20:21], 3", "s_add_u32 s24, s16, s22", "s_addc_u32 s25, s17, s23", "v_mov_b32_e32 v6, 0",
             "global_store_dwordx2 v6, v[2:3], s[24:25]", "s_load_dwordx2 s[12:13], s[16:17], 0x0"]
    found = verdict(head=after)
    assert len(found) == 1 and "can execute AFTER the vector store" in found[0], found
    # SGPRs spilled to VGPR lanes keep their argument (v_writelane / v_readlane)
    spill = ["s_load_dwordx2 s[16:17], s[0:1], 0x0", "s_waitcnt lgkmcnt(0)", "v_writelane_b32 v40, s16, 0", "v_writelane_b32 v40, s17, 1",
             "s_mov_b64 s[16:17], 0", "v_readlane_b32 s30, v40, 0", "v_readlane_b32 s31, v40, 1", "v_mov_b32_e32 v6, 0",
             "global_store_dword v6, v7, s[30:31]", "s_load_dword s33, s[30:31], 0x0"]
    found = verdict(head=spill)
    assert len(found) == 1 and "argument +0x0" in found[0], found
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc here: the current tree cannot be compiled to ISA")
    now = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=900)
    assert now.returncode == 0 and " 0 candidate hazard(s)" in now.stdout, now.stdout[-3000:] + now.stderr[-2000:]
    m = re.search(r"(\d+) of (\d+) scalar data loads, (\d+) of (\d+) vector stores", now.stdout)
    assert m and m.group(1) == m.group(2) and m.group(3) == m.group(4) and int(m.group(2)) > 100 and int(m.group(4)) > 1000, now.stdout[-600:]


def test_hazard_lint_catches_the_pattern_compiled_from_source(tmp_path):
    """Mutation test of the lint's whole pipeline: tests/hazard_mutation.hip holds round 3's faulty
    ``k_colscale_after_truncate`` and its fixed form as SOURCE; compiled with today's hipcc, the first must be flagged
    (the compiler still picks a scalar load and waits for it behind the store loop) and the second must pass."""
    import subprocess
    import sys

    from conftest import ROOT

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    out = os.path.join(tmp_path, "mut.s")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out,
                        os.path.join(ROOT, "tests", "hazard_mutation.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lint = subprocess.run([sys.executable, os.path.join(ROOT, "arnoldi-py_amd", "csrc", "check_scalar_hazards.py"), out],
                          capture_output=True, text=True)
    assert lint.returncode == 1, lint.stdout
    flagged = [ln for ln in lint.stdout.splitlines() if ": line " in ln]
    assert flagged and all(ln.startswith("k_colscale_r03") for ln in flagged), lint.stdout
    assert "can overtake it" in lint.stdout and "k_colscale_fixed: line" not in lint.stdout


def test_complex_schur_takes_the_real_route_only_when_it_is_a_complex_schur_form(monkeypatch):
    """utils.complex_schur: for an exactly real matrix with a real spectrum the real Schur form (dgees, a third of the
    time of zgees) is triangular and is returned as the complex Schur form; 2 x 2 blocks or any imaginary part mean zgees."""
    import scipy.linalg
    from arnoldi_amd.utils import complex_schur

    rng = np.random.default_rng(0)
    S = rng.standard_normal((12, 12))
    sym = (S + S.T).astype(np.complex128)                       # real spectrum
    T, Z = complex_schur(sym)
    assert T.dtype == Z.dtype == np.complex128 and not T.imag.any() and not Z.imag.any()
    assert np.abs(np.tril(T, -1)).max() == 0.0
    np.testing.assert_allclose(Z @ T @ Z.conj().T, sym, atol=1e-12)
    np.testing.assert_allclose(Z.conj().T @ Z, np.eye(12), atol=1e-13)
    np.testing.assert_allclose(np.sort(np.diag(T).real), np.linalg.eigvalsh(sym.real), rtol=1e-12)
    rot = S.astype(np.complex128)                               # real, complex pairs: the real form has 2 x 2 blocks
    Tz, Zz = scipy.linalg.schur(rot, output="complex")
    T2, Z2 = complex_schur(rot)
    np.testing.assert_array_equal(T2, Tz)
    np.testing.assert_array_equal(Z2, Zz)
    cplx = sym + 1e-3j * S                                      # not real: zgees
    np.testing.assert_array_equal(complex_schur(cplx)[0], scipy.linalg.schur(cplx, output="complex")[0])
    # a solver's memo: after a real attempt that met 2 x 2 blocks the next 7 calls go straight to zgees, the 8th tries again
    calls = []
    real_schur = scipy.linalg.schur
    monkeypatch.setattr(scipy.linalg, "schur", lambda a, output="real", **kw: (calls.append(output), real_schur(a, output=output, **kw))[1])
    memo = {}
    for _ in range(9):
        np.testing.assert_array_equal(complex_schur(rot, memo)[0], Tz)
    assert calls.count("real") == 2 and calls.count("complex") == 9 and memo["real_attempts_wasted"] == 2
    calls.clear()
    memo = {}
    for _ in range(3):                                         # a real spectrum never arms the memo
        complex_schur(sym, memo)
    assert calls == ["real"] * 3 and not memo
    monkeypatch.undo()
    monkeypatch.setenv("AKS_REAL_SCHUR", "0")
    np.testing.assert_array_equal(complex_schur(sym)[0], scipy.linalg.schur(sym, output="complex")[0])
