"""Checks of the explicit-restart path shared by the CPU tests (host logic over tests/fake_hip.py)
and the GPU parity tests (real kernels): the reference's own tests (tests/test_explicit_restarts.py)
restated, the golden vectors G10 (reference outputs) and the CPU oracle on the same seeds."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from conftest import load_golden

C128 = np.complex128


def laplace2d(nx, ny):
    Lx, Ly = sp.csr_matrix(oracle.laplace_1d(nx)), sp.csr_matrix(oracle.laplace_1d(ny))
    return (sp.kron(sp.eye(ny), Lx) + sp.kron(Ly, sp.eye(nx))).tocsr()


def assert_same_directions(vecs, ref, atol):
    """Eigenvector columns agree up to a unit complex factor (LAPACK's normalisation of the small
    eigenproblem's vectors depends on rounding-level differences in H)."""
    for j in range(ref.shape[1]):
        phase = np.vdot(ref[:, j], vecs[:, j])
        assert abs(abs(phase) - 1.0) < atol, (j, abs(phase))
        np.testing.assert_allclose(vecs[:, j], ref[:, j] * (phase / abs(phase)), rtol=0, atol=atol)


def check_ritz_decomposition():
    from arnoldi_amd.decomposition import RitzDecomposition
    from arnoldi_amd.utils import arg_largest_real

    g = load_golden("g10_explicit_restarts")
    A = oracle.mark_matrix(10)
    for tag, nr, fn in (("lm3", 3, None), ("lr8", 8, arg_largest_real)):
        r = RitzDecomposition.from_v_and_h(g["ritz_V"], g["ritz_H"], nr, sort_function=fn)
        np.testing.assert_allclose(r.values, g[f"ritz_{tag}_values"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(r.approximate_residuals, g[f"ritz_{tag}_approx"], rtol=1e-12, atol=1e-13)
        assert r.vectors.shape == (A.shape[0], nr)
        np.testing.assert_allclose(r.vectors, g[f"ritz_{tag}_vectors"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(r.compute_true_residuals(A), g[f"ritz_{tag}_true"], rtol=1e-10)
    # a decomposition built from host vectors (the dataclass constructor of the reference)
    r2 = RitzDecomposition(g["ritz_lm3_values"], g["ritz_lm3_vectors"], g["ritz_lm3_approx"])
    np.testing.assert_allclose(r2.compute_true_residuals(A), g["ritz_lm3_true"], rtol=1e-10)
    with pytest.raises(AssertionError):
        RitzDecomposition.from_v_and_h(g["ritz_V"], g["ritz_H"], 9)      # n_ritz > max_dim


def check_ritz_wide(n=3000, m=100, q=100):
    """More Ritz vectors than one launch holds (column chunks), against NumPy."""
    from arnoldi_amd.decomposition import RitzDecomposition

    rng = np.random.default_rng(5)
    V, _ = np.linalg.qr(rng.standard_normal((n, m + 1)) + 1j * rng.standard_normal((n, m + 1)))
    H = np.triu(rng.standard_normal((m + 1, m)) + 1j * rng.standard_normal((m + 1, m)), -1)
    r = RitzDecomposition.from_v_and_h(V, H, q)
    ro = oracle.ritz_from_v_and_h(V, H, q)
    np.testing.assert_allclose(r.values, ro.values, rtol=1e-13)
    np.testing.assert_allclose(r.vectors, ro.vectors, rtol=0, atol=1e-12)


def check_mgs():
    from arnoldi_amd.explicit_restarts import mgs

    g = load_golden("g10_explicit_restarts")
    for k in (0, 1, 6):
        w = g["mgs_w_in"].copy()
        out = mgs(g["mgs_basis"][:, :k], w, 1e-8)
        assert out is w
        np.testing.assert_allclose(w, g[f"mgs_w_out_{k}"], rtol=0, atol=1e-14)
    with pytest.raises(AssertionError, match="Too small norm"):
        mgs(g["mgs_basis"][:, :2], g["mgs_basis"][:, 0].copy(), 1e-8)


def check_naive():
    from arnoldi_amd.explicit_restarts import naive_explicit_restarts

    g = load_golden("g10_explicit_restarts")
    A = oracle.mark_matrix(10)
    for restarts, digits in [(1, 0), (2, 1), (3, 3), (4, 5), (5, 6)]:     # Saad table 6.2
        np.random.seed(0)
        ritz, ok, used = naive_explicit_restarts(A, 10, max_restarts=restarts)
        res = ritz.compute_true_residuals(A)
        assert res <= 2 * 10.0 ** (-digits)
        np.testing.assert_allclose(ritz.values, g[f"naive_r{restarts}_value"], rtol=1e-10)
        np.testing.assert_allclose(res, g[f"naive_r{restarts}_true"], rtol=1e-4)
        assert [int(ok), used] == list(g[f"naive_r{restarts}_flags"])
    np.random.seed(0)
    ritz, ok, used = naive_explicit_restarts(A, 20, max_restarts=200, stopping_criterion=1e-6)
    assert ok and [int(ok), used] == list(g["naive_conv_flags"])
    assert ritz.compute_true_residuals(A) <= 1e-6
    np.testing.assert_allclose(ritz.values, g["naive_conv_value"], rtol=1e-10)
    assert ritz.vectors.shape == (A.shape[0], 1)
    np.testing.assert_allclose(ritz.vectors[:, 0], g["naive_conv_vector"], rtol=0, atol=1e-9)


DEFLATION_CASES = {
    "defl_mark10": lambda g: (oracle.mark_matrix(10), 3, 0,
                              dict(max_dim=10, stopping_criterion=1e-8, sort_function=oracle.arg_largest_real)),
    "defl_diag": lambda g: (g["defl_diag_A"], 3, 0, {}),
    "defl_mark30": lambda g: (oracle.mark_matrix(30), 4, 1,
                              dict(max_dim=30, stopping_criterion=1e-8, sort_function=oracle.arg_largest_real)),
    "defl_lap": lambda g: (laplace2d(12, 13), 3, 2, dict(max_dim=30, stopping_criterion=1e-6, max_restarts=400)),
}


def check_deflation(tag):
    from arnoldi_amd.explicit_restarts import explicit_restarts_with_deflation

    g = load_golden("g10_explicit_restarts")
    M, nev, seed, kw = DEFLATION_CASES[tag](g)
    np.random.seed(seed)
    stats = {}
    vals, vecs, hist = explicit_restarts_with_deflation(M, nev, stats=stats, **kw)
    assert hist.k == nev and vecs.shape == (M.shape[0], nev)
    np.testing.assert_array_equal(hist.matvecs, g[f"{tag}_matvecs"])
    np.testing.assert_array_equal(hist.restarts, g[f"{tag}_restarts"])
    np.testing.assert_allclose(vals, g[f"{tag}_vals"], rtol=1e-9, atol=1e-12)
    res = np.linalg.norm(M @ vecs - vals * vecs, axis=0)
    assert np.all(res <= np.maximum(2.0 * g[f"{tag}_residuals"], 1e-12)), (res, g[f"{tag}_residuals"])
    if tag != "defl_diag":      # (a double eigenvalue's vectors span a plane: residuals only)
        assert_same_directions(vecs, g[f"{tag}_vecs"], 1e-6)
    # book-keeping: true operator applications = Arnoldi steps + one per converged value
    assert stats["matvecs"] >= int(hist.matvecs.sum())


def check_deflation_reference_tests():
    """tests/test_explicit_restarts.py:84-160: values against ARPACK through the Hungarian matching,
    the default-argument path and the failure message."""
    from scipy.optimize import linear_sum_assignment
    from scipy.sparse.linalg import eigs

    from arnoldi_amd.explicit_restarts import explicit_restarts_with_deflation
    from arnoldi_amd.utils import arg_largest_real

    def against_arpack(A, k, max_dim=None, which="LM", tol=None, max_restarts=100):
        sort_function = None if which == "LM" else arg_largest_real
        r_vals = eigs(A, k, which=which)[0]
        vals, vecs, history = explicit_restarts_with_deflation(
            A, k, max_dim=max_dim, stopping_criterion=tol, sort_function=sort_function, max_restarts=max_restarts)
        residuals = np.linalg.norm(A @ vecs - vals * vecs, axis=0)
        assert history.k == k
        np.testing.assert_allclose(residuals, 0, rtol=1e-4, atol=1e-08)
        cost = np.abs(vals[:, None] - r_vals[None, :])
        ri, ci = linear_sum_assignment(cost)
        np.testing.assert_allclose(vals[ri], r_vals[ci], rtol=1e-4, atol=1e-08)

    np.random.seed(4)
    against_arpack(oracle.mark_matrix(10), 3, 10, which="LR", tol=1e-8)          # Saad table 6.3
    D = np.diag([7.0, 7, 5, 4, 3, 2, 1])
    Q, _ = np.linalg.qr(np.random.randn(7, 7))
    against_arpack(Q.T @ D @ Q, 3)
    with pytest.raises(ValueError, match="Could not converge for value 0"):
        against_arpack(oracle.mark_matrix(10), 3, max_dim=5, tol=1e-16, max_restarts=10)


def check_happy_breakdown_deflate():
    """partial_schur(on_breakdown="deflate"): a start vector inside a 6-dimensional invariant subspace
    stops the expansion at m = 6 < max_dim; the wanted eigenvalues of that block come back exactly.
    The default still raises like the reference (krylov_schur.py:57-59)."""
    import arnoldi_amd
    from arnoldi_amd.utils import arg_largest_real

    rng = np.random.default_rng(9)
    B = rng.standard_normal((6, 6))
    Cc = sp.random(200, 200, density=0.03, random_state=np.random.RandomState(1)) + sp.eye(200) * 3
    A = sp.block_diag([sp.csr_matrix(B), Cc], format="csr")
    v0 = np.zeros(206, C128)
    v0[:6] = rng.standard_normal(6)
    v0 /= np.linalg.norm(v0)
    with pytest.raises(ValueError, match="Happy breakdown not supported yet"):
        arnoldi_amd.partial_schur(A, 3, max_dim=12, v0=v0)
    st = {}
    Q, T, hist = arnoldi_amd.partial_schur(A, 3, max_dim=12, v0=v0, on_breakdown="deflate",
                                           sort_function=arg_largest_real, stats=st)
    want = np.linalg.eigvals(B)
    want = want[np.argsort(-want.real)][:3]
    got = np.diag(T)
    assert np.abs(got[:, None] - want[None, :]).min(axis=1).max() < 1e-10       # same set (order of a
    assert np.abs(got[:, None] - want[None, :]).min(axis=0).max() < 1e-10       # conjugate pair is free)
    np.testing.assert_allclose(A @ Q, Q @ T, atol=1e-10)
    np.testing.assert_allclose(Q.conj().T @ Q, np.eye(3), atol=1e-12)
    assert st["restarts"] == 1 and list(hist.restarts) == [1, 1, 1]
    with pytest.raises(ValueError, match="invariant subspace of dimension 6 < nev"):
        arnoldi_amd.partial_schur(A, 8, max_dim=20, v0=v0, on_breakdown="deflate")


def check_ritz_reference_tests():
    """TestRitzDecomposition of the reference (tests/test_decomposition.py:174-261) restated on the
    device-backed class: overlap with ARPACK's vectors, true vs approximate residuals, ``max_dim``."""
    from scipy.sparse.linalg import eigs

    from arnoldi_amd.decomposition import RitzDecomposition, arnoldi_decomposition
    from arnoldi_amd.matrices import laplace, mark
    from arnoldi_amd.utils import rand_normalized_vector

    RTOL, ATOL = 1e-4, 1e-8

    def factorise(A, m):
        n = A.shape[0]
        V = np.zeros((n, m + 1), dtype=C128)
        H = np.zeros((m + 1, m), dtype=C128)
        V[:, 0] = rand_normalized_vector(n, C128)
        V, H, n_iter = arnoldi_decomposition(A, V, H)
        return V, H

    np.random.seed(11)
    A = mark(10)
    for which, sort_function in (("LM", lambda x: np.argsort(-np.abs(x))), ("LR", lambda x: np.argsort(-np.real(x)))):
        r_vecs = eigs(A, 2, which=which)[1]                                    # test_simple
        V, H = factorise(A, 30)
        ritz = RitzDecomposition.from_v_and_h(V, H, 2, sort_function=sort_function)
        overlap = np.linalg.norm(ritz.vectors.T @ r_vecs) / np.sqrt(2)
        np.testing.assert_allclose(overlap, 1, rtol=1e-4, atol=ATOL)
        assert np.linalg.norm(A @ ritz.vectors - ritz.values * ritz.vectors) <= 2e-3
    for M, m in ((mark(10), 20), (laplace(100), 10)):                          # test_residual_computation
        V, H = factorise(M, m)
        ritz = RitzDecomposition.from_v_and_h(V, H, 2)
        residuals = np.linalg.norm(M @ ritz.vectors - ritz.values * ritz.vectors, axis=0)
        np.testing.assert_allclose(ritz.compute_true_residuals(M), residuals, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(ritz.approximate_residuals, residuals, rtol=RTOL, atol=ATOL)
    m, max_dim = 20, 15                                                         # test_max_dim
    V, H = factorise(A, m)
    V, H = np.array(V), np.array(H)
    V[:, max_dim:] = np.random.randn(*V[:, max_dim:].shape)
    H[max_dim + 1:, max_dim:] = np.random.randn(*H[max_dim + 1:, max_dim:].shape)
    broken = RitzDecomposition.from_v_and_h(V, H, 2)
    good = RitzDecomposition.from_v_and_h(V, H, 2, max_dim=max_dim)
    with pytest.raises(AssertionError):
        np.testing.assert_allclose(broken.compute_true_residuals(A), broken.approximate_residuals, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(good.compute_true_residuals(A), good.approximate_residuals, rtol=RTOL, atol=ATOL)
