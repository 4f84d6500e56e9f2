"""Checks of the real-arithmetic mode (partial_schur(arithmetic="real")) shared by the CPU tests (host
logic over tests/fake_hip.py) and the GPU tests (real kernels).  The iteration differs from the
reference's complex one where a restart would cut a conjugate pair, so the comparison is the north
star's: the same eigenvalues as the CPU oracle of the reference's algorithm, residuals
max ||A v - l v|| / |l| no worse than 1.05 x the oracle's (floor 1e-12), a valid partial Schur pair."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle

C128 = np.complex128


def _match(a, b):
    """max over a of the distance to the closest b, both directions (sets of eigenvalues)."""
    fold = lambda z: np.asarray(z).real + 1j * np.abs(np.asarray(z).imag)   # noqa: E731  (which member of a
    a, b = fold(a), fold(b)                                    # conjugate pair is cut off at nev is arbitrary)
    d = np.abs(a[:, None] - b[None, :])
    return max(d.min(axis=1).max(), d.min(axis=0).max())


def solve_both(A, nev, seed, **kw):
    import arnoldi_amd

    np.random.seed(seed)
    st = {}
    Q, T, hist = arnoldi_amd.partial_schur(A, nev, arithmetic="real", stats=st, **kw)
    okw = dict(kw)
    np.random.seed(seed)
    Qo, To, histo = oracle.krylov_schur(A, nev, **okw)
    return (Q, T, hist, st), (Qo, To, histo)


def check_case(A, nev, seed, **kw):
    (Q, T, hist, st), (Qo, To, histo) = solve_both(A, nev, seed, **kw)
    n = A.shape[0]
    assert Q.shape == (n, nev) and T.shape == (nev, nev) and Q.dtype == C128 and T.dtype == C128
    assert st["arithmetic"] == "real"
    assert np.abs(np.tril(T, -1)).max() == 0.0                         # upper triangular, like the reference's T
    np.testing.assert_allclose(Q.conj().T @ Q, np.eye(nev), atol=1e-11)
    eig_tol = 10 * float(st["tol"])                 # both iterations stop at residual ~ tol
    assert _match(np.diag(T), np.diag(To)) < eig_tol * max(1.0, np.abs(np.diag(To)).max()), (np.diag(T), np.diag(To))
    _, _, rel = oracle.eig_residuals(A, Q, T)
    _, _, rel_o = oracle.eig_residuals(A, Qo, To)
    assert rel.max() <= max(1.05 * rel_o.max(), 10 * float(st["tol"])), (rel, rel_o)
    scale = np.abs(np.diag(T)).max()
    assert np.linalg.norm(A @ Q - Q @ T, axis=0).max() <= 50 * float(st["tol"]) * scale
    # residuals of the real partial Schur form evaluated on the device (real-packed kernels: two real operator
    # applications + two J = 2 fused updates per eigenvector) against the same quantity on the host
    solver = st["solver"]
    k = solver.nev_now
    dvals, dres, drel = solver.true_residuals()
    Qr = solver.ctx.gather_columns(0, k)
    hv, hS = np.linalg.eig(solver.H[:k, :k])
    hres = np.linalg.norm(A @ (Qr @ hS) - (Qr @ hS) * hv, axis=0)
    np.testing.assert_allclose(dvals, hv, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(dres, hres, rtol=1e-6, atol=1e-13 * max(scale, 1.0))
    # restart counts are close to the complex iteration's (not necessarily equal)
    assert abs(int(st["restarts"]) - int(histo.restarts.max())) <= max(3, int(0.3 * histo.restarts.max()))
    return st


def planted_pairs(n, seed=3):
    from arnoldi_amd import matrices

    A = (0.25 * matrices.random_csr(n, 5, seed)).tolil()
    for k, (a, b) in enumerate([(3.5, 2.0), (3.0, 1.0), (2.0, 2.2)]):
        i = 17 + 400 * k
        A[i, i], A[i, i + 1], A[i + 1, i], A[i + 1, i + 1] = a, b, -b, a
    return A.tocsr()


def cases():
    from arnoldi_amd import matrices

    LR, LM = oracle.arg_largest_real, oracle.arg_largest_magnitude
    rng = np.random.default_rng(4)
    dense = rng.standard_normal((90, 90))
    return {
        "mark30_lr": (matrices.mark(30), 4, 0, dict(max_dim=20, stopping_criterion=1e-8, sort_function=LR)),
        "mark50_readme": (matrices.mark(50), 5, 1, dict(max_dim=20, stopping_criterion=1e-8, sort_function=LR)),
        "planted_odd_n": (matrices.random_csr(6001, 5, 1234, planted=(4.0, 3.7, 3.4, 3.1, 2.8, 2.5)), 5, 0,
                          dict(max_dim=20, sort_function=LM)),
        "laplace2d": (matrices.laplace2d(30, 31), 10, 0, dict(max_dim=40, sort_function=LM)),
        "conjugate_pairs": (matrices.random_csr(3000, 5, 7), 6, 0,
                            dict(max_dim=30, stopping_criterion=1e-6, max_restarts=3000, sort_function=LM)),
        # well separated planted pairs 3.5 +- 2i, 3 +- 1i, 2 +- 2.2i over a small random background: nev = 3 and
        # nev = 5 cut a conjugate pair at the boundary of the wanted set, p = nev + 5 cuts one at the restart
        "pair_cut_at_nev3": (planted_pairs(2500), 3, 2, dict(max_dim=24, stopping_criterion=1e-9, sort_function=LM)),
        "pair_cut_at_nev5": (planted_pairs(2501), 5, 5, dict(max_dim=20, stopping_criterion=1e-9, sort_function=LM)),
        "dense_array": (dense, 4, 3, dict(max_dim=30, stopping_criterion=1e-9, max_restarts=2000, sort_function=LM)),
    }


def check_errors():
    import arnoldi_amd
    from arnoldi_amd import matrices

    A = matrices.mark(10)
    with pytest.raises(ValueError, match="real matrix"):
        arnoldi_amd.partial_schur(A.astype(C128), 2, arithmetic="real")
    with pytest.raises(ValueError, match="real start vector"):
        arnoldi_amd.partial_schur(A, 2, arithmetic="real", v0=np.full(A.shape[0], 1 + 1j))
    with pytest.raises(AssertionError):
        arnoldi_amd.partial_schur(A, 2, arithmetic="quaternion")
    with pytest.raises(ValueError, match="max_dim >= nev"):
        arnoldi_amd.partial_schur(A, 3, max_dim=4, arithmetic="real")
    np.random.seed(0)
    with pytest.raises(ValueError, match="Has not converged"):
        arnoldi_amd.partial_schur(matrices.random_csr(2000, 5, 1234), 5, max_dim=20, max_restarts=3,
                                  arithmetic="real")
    # a complex-typed start vector with zero imaginary part is fine (that is what the reference draws)
    v0 = np.random.default_rng(0).standard_normal(A.shape[0]).astype(C128)
    Q, T, _ = arnoldi_amd.partial_schur(A, 2, arithmetic="real", v0=v0 / np.linalg.norm(v0),
                                        sort_function=oracle.arg_largest_real, max_dim=12)
    assert abs(T[0, 0] - 1.0) < 1e-8


def check_auto():
    """arithmetic="auto": real for a real matrix + real start vector, complex otherwise."""
    import arnoldi_amd
    from arnoldi_amd import matrices

    A = matrices.mark(12)
    kw = dict(max_dim=12, sort_function=oracle.arg_largest_real, stopping_criterion=1e-9)
    for M, v0, want in ((A, None, "real"), (A.astype(C128), None, "complex"),
                        (A, np.exp(1j * np.arange(A.shape[0])) / np.sqrt(A.shape[0]), "complex"),
                        (A.toarray(), None, "real")):
        st = {}
        np.random.seed(5)
        Q, T, _ = arnoldi_amd.partial_schur(M, 2, arithmetic="auto", v0=v0, stats=st, **kw)
        assert st["arithmetic"] == want and abs(T[0, 0] - 1.0) < 1e-8


def check_reorder_real_schur():
    """Block reordering with dtrexc: every sort key, random quasi-triangular inputs with pairs."""
    import scipy.linalg as sla
    from arnoldi_amd.krylov_schur_real import block_eigenvalues, real_blocks, reorder_real_schur

    rng = np.random.default_rng(12)
    for trial in range(20):
        m = int(rng.integers(2, 25))
        H = np.triu(rng.standard_normal((m, m)), -1)
        T0, Z0 = sla.schur(H, output="real")
        for key in (oracle.arg_largest_magnitude, oracle.arg_largest_real):
            T, Z = reorder_real_schur(T0.copy(), Z0.copy(), key)
            np.testing.assert_allclose(Z @ T @ Z.T, H, atol=1e-11)
            np.testing.assert_allclose(Z.T @ Z, np.eye(m), atol=1e-12)
            assert np.abs(np.tril(T, -2)).max() == 0.0
            ev = block_eigenvalues(T)
            assert _match(ev, np.linalg.eigvals(H)) < 1e-9
            # block keys are non-increasing in quality: each block's best member ranks after the previous block's
            score = np.abs(ev) if key is oracle.arg_largest_magnitude else ev.real
            best = [max(score[s: s + size]) for s, size in real_blocks(T)]
            assert all(best[i] >= best[i + 1] - 1e-9 for i in range(len(best) - 1)), best


# ---- the real-arithmetic counterparts of the widened solvers ("real-arithmetic islands") ----------------
def check_locking_real(A, nev, seed, **kw):
    """partial_schur(arithmetic="real", locking=True) against the oracle of the reference's algorithm."""
    import arnoldi_amd

    np.random.seed(seed)
    st = {}
    Q, T, hist = arnoldi_amd.partial_schur(A, nev, arithmetic="real", locking=True, stats=st, max_restarts=3000, **kw)
    np.random.seed(seed)
    Qo, To, histo = oracle.krylov_schur(A, nev, max_restarts=3000, **kw)
    tol = float(st["tol"])
    assert st["arithmetic"] == "real" and st["locked"] >= nev and Q.shape == (A.shape[0], nev)
    assert _match(np.diag(T), np.diag(To)) < 10 * tol * max(1.0, np.abs(np.diag(To)).max()), (np.diag(T), np.diag(To))
    assert np.abs(np.tril(T, -1)).max() == 0.0
    np.testing.assert_allclose(Q.conj().T @ Q, np.eye(nev), atol=1e-11)
    _, _, rel = oracle.eig_residuals(A, Q, T)
    _, _, rel_o = oracle.eig_residuals(A, Qo, To)
    assert rel.max() <= max(1.05 * rel_o.max(), 10 * tol), (rel.max(), rel_o.max())
    tb = st["truncation_bytes"]
    assert min(tb) < tb[0] or st["restarts"] <= 2              # locked columns leave the restart compression
    assert 0 < st["restarts"] <= 2 * int(histo.restarts.max()) + 5
    return st


def check_deflate_real():
    """on_breakdown="deflate" in real arithmetic: a start vector inside a 7-dimensional invariant subspace (a
    diagonal block holding a conjugate pair) => the expansion breaks down at m = 7 and the wanted Schur vectors of
    that subspace come back, as from the complex driver."""
    import arnoldi_amd

    rng = np.random.default_rng(11)
    B = np.diag([5.0, 4.0, 3.0, 2.5, 1.0]).astype(float)
    B = np.block([[B, np.zeros((5, 2))], [np.zeros((2, 5)), np.array([[4.5, 1.5], [-1.5, 4.5]])]])
    R, _ = np.linalg.qr(rng.standard_normal((7, 7)))
    n = 60
    A = np.zeros((n, n))
    A[:7, :7] = R @ B @ R.T
    A[7:, 7:] = np.diag(np.linspace(0.1, 0.9, n - 7))
    v0 = np.zeros(n)
    v0[:7] = rng.standard_normal(7)
    v0 /= np.linalg.norm(v0)
    out = {}
    for mode in ("complex", "real"):
        st = {}
        Q, T, hist = arnoldi_amd.partial_schur(sp.csr_matrix(A), 3, max_dim=20, v0=v0.copy(), on_breakdown="deflate",
                                               arithmetic=mode, stats=st, sort_function=oracle.arg_largest_magnitude)
        assert st["restarts"] == 1 and Q.shape == (n, 3)
        assert np.linalg.norm(A @ Q - Q @ T) < 1e-10 and np.abs(Q.conj().T @ Q - np.eye(3)).max() < 1e-11
        out[mode] = np.diag(T)
    want = np.array([5.0, 4.5 + 1.5j, 4.5 - 1.5j])           # |4.5 +- 1.5i| = 4.74 < 5
    assert _match(out["real"], want) < 1e-9 and _match(out["complex"], want) < 1e-9
    with pytest.raises(ValueError, match="Happy breakdown not supported yet"):
        arnoldi_amd.partial_schur(sp.csr_matrix(A), 3, max_dim=20, v0=v0.copy(), arithmetic="real")


def check_residual_norms_real():
    """ArnoldiContext.residual_norms on a real-packed basis: real eigenpairs directly, complex ones refused with a
    pointer to the pair routine, whose result matches the host's."""
    import arnoldi_amd
    from arnoldi_amd import matrices

    A = matrices.mark(30)
    np.random.seed(3)
    st = {}
    arnoldi_amd.partial_schur(A, 4, arithmetic="real", max_dim=20, stopping_criterion=1e-9, stats=st,
                              sort_function=oracle.arg_largest_real)
    solver = st["solver"]
    ctx, k = solver.ctx, solver.nev_now
    vals, S = np.linalg.eig(solver.H[:k, :k])
    assert not vals.imag.any()                                # the dominant eigenvalues of the Markov chain are real
    block = ctx.combine(0, k, S.real)
    got = ctx.residual_norms(block, vals.real)
    Qr = ctx.gather_columns(0, k)
    U = Qr @ S.real
    np.testing.assert_allclose(got, np.linalg.norm(A @ U - U * vals.real, axis=0), rtol=1e-6, atol=1e-14)
    with pytest.raises(ValueError, match="values must be real"):
        ctx.residual_norms(block, vals + 1e-3j)
    np.testing.assert_allclose(ctx.residual_norms_real_pairs(k, S, vals), got, rtol=1e-6, atol=1e-14)


def check_explicit_deflation_real(A, nev, seed, **kw):
    """explicit_restarts_with_deflation(arithmetic="real") against the oracle of the reference's (complex) solver."""
    from arnoldi_amd.explicit_restarts import explicit_restarts_with_deflation

    np.random.seed(seed)
    st = {}
    vals, vecs, hist = explicit_restarts_with_deflation(A, nev, arithmetic="real", stats=st, **kw)
    np.random.seed(seed)
    vo, xo, ho = oracle.explicit_restarts_with_deflation(A, nev, **kw)
    tol = float(st["tol"])
    assert vals.shape == (nev,) and vecs.shape == (A.shape[0], nev) and st["locked"] in (nev, nev + 1)
    assert _match(vals, vo) < 10 * tol * max(1.0, np.abs(vo).max()), (vals, vo)
    res = np.linalg.norm(A @ vecs - vecs * vals, axis=0) / np.abs(vals)
    res_o = np.linalg.norm(A @ xo - xo * vo, axis=0) / np.abs(vo)
    assert res.max() <= max(1.05 * res_o.max(), 10 * tol), (res, res_o)
    K, ctx = st["locked"], st["ctx"]
    Hk = st["H"][:K, :K]
    assert np.abs(np.tril(Hk, -2)).max() == 0.0               # real quasi-triangular
    Qr = ctx.gather_columns(0, K)
    np.testing.assert_allclose(Qr.T @ Qr, np.eye(K), atol=1e-10)
    # device-side residuals of the (complex) eigenpairs of the locked block against the host's
    ev, Y = np.linalg.eig(Hk)
    dres = ctx.residual_norms_real_pairs(K, Y, ev)
    U = Qr @ Y
    np.testing.assert_allclose(dres, np.linalg.norm(A @ U - U * ev, axis=0), rtol=1e-5, atol=1e-13)
    assert np.all(hist.restarts > 0)
    return st
