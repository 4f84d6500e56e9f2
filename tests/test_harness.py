"""Harness counterpart: loaders, eigenvalue matching, ARPACK comparison."""
import os

import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp

import oracle

C128 = np.complex128


def test_load_matrix_formats(tmp_path):
    from arnoldi_amd import harness, matrices

    A = matrices.mark(12)
    mat = os.path.join(tmp_path, "m.mat")
    # SuiteSparse layout: struct Problem with field A
    scipy.io.savemat(mat, {"Problem": {"A": sp.csc_matrix(A), "name": "mark12"}})
    mtx = os.path.join(tmp_path, "m.mtx")
    scipy.io.mmwrite(mtx, sp.coo_matrix(A))
    npz = os.path.join(tmp_path, "m.npz")
    sp.save_npz(npz, sp.coo_matrix(A))
    plain = os.path.join(tmp_path, "plain.mat")
    scipy.io.savemat(plain, {"M": sp.csc_matrix(A)})
    for path in (mat, mtx, npz, plain):
        B = harness.load_matrix(path)
        assert sp.isspmatrix_csr(B) and B.has_sorted_indices and (B - A).nnz == 0
    with pytest.raises(ValueError, match="unknown matrix file type"):
        harness.load_matrix(os.path.join(tmp_path, "x.txt"))
    rect = os.path.join(tmp_path, "r.npz")
    sp.save_npz(rect, sp.random(4, 5, 0.5, format="coo"))
    with pytest.raises(ValueError, match="not square"):
        harness.load_matrix(rect)


def test_load_matrix_matlab_v73(tmp_path):
    """MATLAB v7.3 = HDF5: recognised by its header; without h5py the error says so, with an h5py-like module
    the CSC arrays of ``Problem/A`` (data, ir, jc + MATLAB_sparse) are assembled."""
    from arnoldi_amd import harness, matrices

    path = os.path.join(tmp_path, "big.mat")
    with open(path, "wb") as f:
        f.write(b"MATLAB 7.3 MAT-file, Platform: GLNXA64, Created on: Thu Jan  1 00:00:00 1970 HDF5 schema 1.00 .".ljust(512))
    assert harness._is_matlab_v73(path)
    try:
        import h5py  # noqa: F401
        have = True
    except ImportError:
        have = False
    if not have:
        with pytest.raises(ValueError, match="h5py is not installed"):
            harness.load_matrix(path)

    A = sp.csc_matrix(matrices.mark(9))

    class Node(dict):
        attrs = {}

    class Sparse(dict):
        def __init__(self, M):
            super().__init__(data=M.data, ir=M.indices, jc=M.indptr)
            self.attrs = {"MATLAB_sparse": M.shape[0]}

    class File(Node):
        def __init__(self, name, mode):
            super().__init__({"#refs#": Node(), "Problem": Node(name=Node(), A=Sparse(A))})

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    class FakeH5:
        pass

    FakeH5.File = File
    B = harness._load_mat_v73(path, h5py=FakeH5)
    assert (B - A).nnz == 0 and B.shape == A.shape


@pytest.mark.gpu
def test_loaded_files_solve_on_the_device(tmp_path):
    """a matrix file -> ``load_matrix`` -> ``partial_schur`` on the MI355X -> the oracle's answer.
    The reference's scripts do exactly this with SuiteSparse ``.mat`` files (scripts/utils.py:102-116,
    benchmark-partial-schur.py:74-100); none can be downloaded here, so the files are written first: a banded
    non-symmetric matrix of af_shell10's row density with a planted dominant spectrum, as ``.mat``
    (``Problem.A``), MatrixMarket and ``.npz``."""
    import arnoldi_amd
    from arnoldi_amd import harness, matrices

    n = 40_000
    A = matrices.banded_csr(n, 35, 1234).tolil()
    planted = (9.0, 8.2, 7.4, 6.6, 5.8, 5.0)
    for i, v in enumerate(planted):
        A[137 * (i + 1), 137 * (i + 1)] = v * 4
    A = sp.csr_matrix(A)
    files = {"mat": os.path.join(tmp_path, "a.mat"), "mtx": os.path.join(tmp_path, "a.mtx"),
             "npz": os.path.join(tmp_path, "a.npz")}
    scipy.io.savemat(files["mat"], {"Problem": {"A": sp.csc_matrix(A), "name": "planted_banded"}})
    scipy.io.mmwrite(files["mtx"], sp.coo_matrix(A), precision=17)
    sp.save_npz(files["npz"], A)
    np.random.seed(3)
    Qo, To, ho = oracle.krylov_schur(A, 4, max_dim=20, sort_function=oracle.arg_largest_magnitude)
    _, _, rel_o = oracle.eig_residuals(A, Qo, To)
    for kind, path in files.items():
        B = harness.load_matrix(path)
        assert (B - A).nnz == 0, kind
        np.random.seed(3)
        st = {}
        Q, T, h = arnoldi_amd.partial_schur(B, 4, max_dim=20, stats=st)
        np.testing.assert_array_equal(h.restarts, ho.restarts)
        np.testing.assert_allclose(np.diag(T), np.diag(To), rtol=1e-9)
        _, _, rel = oracle.eig_residuals(A, Q, T)
        assert rel.max() <= max(1.05 * rel_o.max(), 1e-13), (kind, rel.max(), rel_o.max())
    # and through the harness entry point the reference's scripts use (eigenpairs sorted by `which`)
    vals, vecs, stats = harness.krylov_schur_eig(harness.load_matrix(files["mat"]),
                                                 harness.EigensolverParameters(4, 20, 1e-8, 1000, None, "LM"))
    assert stats.matvecs > 0 and np.all(np.linalg.norm(A @ vecs - vecs * vals, axis=0) / np.abs(vals) < 5e-8)


def test_find_best_matching_and_residuals():
    from arnoldi_amd import harness

    a = np.array([1 + 1j, 3.0, -2.0, 0.5j])
    b = a[[2, 0, 3, 1]] + 1e-9
    x, y = harness.find_best_matching(a, b)
    np.testing.assert_allclose(x, y, atol=1e-8)
    A = np.diag([2.0, 5.0])
    res, rel = harness.true_residuals(A, np.array([2.0, 5.0]), np.eye(2))
    assert np.all(res == 0) and np.all(rel == 0)
    assert "eigval[1]" in harness.residual_report("x", A, np.array([2.0, 4.0]), np.eye(2))
    assert harness.STRESS_GRID[0] == (3, 20, 10) and len(harness.STRESS_GRID) == 9


def test_compare_and_sweep_on_fake_device(monkeypatch, tmp_path):
    """Orchestration only (device entry points replaced by tests/fake_hip.py)."""
    import fake_hip
    from arnoldi_amd import harness, matrices

    fake_hip.install(monkeypatch)
    A = matrices.mark(30)
    np.random.seed(0)
    rows = harness.compare(A, harness.EigensolverParameters(4, 20, 1e-8, 2000, None, "LR"))
    assert [r["method"] for r in rows] == ["arpack", "krylov-schur-mi355x"]
    assert all(r["match"] for r in rows) and rows[1]["max_rel_residual"] < 5e-8
    assert rows[1]["matvecs"] == 20 + (rows[1]["restarts"] - 1) * 11      # m + R (m - p), p = 9
    out = os.path.join(tmp_path, "sweep.csv")
    np.random.seed(1)
    all_rows = harness.sweep(A, out, grid=[(3, 20, 10), (6, 20, 12)], whichs=("LR",))
    assert len(all_rows) == 4 and all(r["match"] for r in all_rows)
    assert open(out).readline().startswith("method,dtype,nev,ncv,tol,max_restarts,p,which,elapsed,matvecs")


@pytest.mark.gpu
def test_arpack_cross_check_on_gpu():
    """scripts/benchmark-partial-schur.py:97-123 restated: normalised residuals < 5 tol and
    eigenvalues equal ARPACK's to rtol = tol after Hungarian matching."""
    from arnoldi_amd import harness, matrices

    for A, which, nev, ncv in ((matrices.mark(60), "LR", 6, 20), (matrices.laplace2d(60, 67), "LM", 6, 24),
                               (matrices.random_csr(50_000, 5, 7, planted=(4.0, 3.5, 3.0, 2.6)), "LM", 4, 20)):
        np.random.seed(0)
        rows = harness.compare(A.astype(C128), harness.EigensolverParameters(nev, ncv, 1e-8, 5000, None, which))
        assert all(r["match"] for r in rows), rows
        assert rows[1]["max_rel_residual"] < 5e-8
        assert rows[1]["matvecs"] > 0 and rows[1]["restarts"] > 0
