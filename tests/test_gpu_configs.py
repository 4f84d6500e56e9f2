"""BASELINE.json configurations at FULL size on the MI355X (``pytest -m gpu``).

The CPU reference cannot finish these in test time, so parity is established through
size-independent properties and analytic answers:

  C2  2-D Laplace 1000 x 1001 (n ~ 1M), k=10, m=40, LM: solve to convergence; eigenvalues against
      the analytic spectrum, residuals ||Av - lv|| / |l| < 5 tol (the reference scripts' check,
      scripts/benchmark-partial-schur.py:97-100), Schur vectors orthonormal.
  C3  af_shell10 stand-ins (banded: n = 1,508,065, 35 per row in one run; shell: 549 x 549 nodes x 5 unknowns, seven
      5 x 5 blocks per row), k=20 -> m=41, p=25 (defaults):
      planted dominant eigenvalues, solve to convergence, residual check; exercises panel widths
      26..41 (widest fused kernels) and the 41 x 25 truncation.
  C4  3-D Laplace 251 x 252 x 253 (n ~ 16M, V = 10.5 GB): expansion + restarts on one GPU; Arnoldi
      invariants  V^H V = I,  A V_m = V_{m+1} H  after restarts, in complex128 on the device.
  C5  random CSR n = 10M with planted spectrum: solve to convergence, planted eigenvalues found,
      residual check (the un-planted matrix never converges: bench.py times it instead).

Small-size twins of C3 / C5 are compared with the CPU oracle run on the same seed.
"""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
C128 = np.complex128


@pytest.fixture(scope="module")
def amd():
    import torch

    assert torch.cuda.is_available()
    import arnoldi_amd

    return arnoldi_amd


def _residuals(A, Q, T):
    vals, S = np.linalg.eig(T)
    vecs = Q @ S
    return vals, np.linalg.norm(A @ vecs - vecs * vals, axis=0) / np.abs(vals)


def test_config2_laplace2d_1m_converges_to_analytic_spectrum(amd):
    from arnoldi_amd import matrices

    nx, ny = 1000, 1001
    A = matrices.laplace2d(nx, ny)
    np.random.seed(0)
    stats = {}
    Q, T, hist = amd.partial_schur(A, 10, max_dim=40, max_restarts=3000, stats=stats)
    tol = np.sqrt(np.finfo(np.float64).eps)
    vals, rel = _residuals(A, Q, T)
    assert rel.max() < 5 * tol, rel
    analytic = np.sort((matrices.laplace_eigen(nx)[:, None] + matrices.laplace_eigen(ny)[None, :]).ravel())[:10]
    np.testing.assert_allclose(np.sort(vals.real), analytic, rtol=1e-9)
    assert np.abs(vals.imag).max() < 1e-9
    np.testing.assert_allclose(Q.conj().T @ Q, np.eye(10), atol=1e-11)
    assert np.all(hist.restarts == stats["restarts"]) and stats["restarts"] > 50
    assert stats["matvecs"] == 40 + (stats["restarts"] - 1) * 25
    print(f"C2: {stats['restarts']} restarts, max rel residual {rel.max():.2e}, "
          f"second passes {stats['second_passes']} of {stats['matvecs']} steps")


def test_config3_banded_stand_in(amd):
    from arnoldi_amd import matrices

    planted = tuple(60.0 - 1.5 * i for i in range(24))
    # small twin against the oracle (same generator, same seed)
    As = matrices.banded_csr(30_000, 35, 1234, planted=planted)
    np.random.seed(0)
    Qo, To, ho = oracle.krylov_schur(As, 20)
    np.random.seed(0)
    st = {}
    Q, T, h = amd.partial_schur(As, 20, stats=st)
    assert (st["max_dim"], st["p"]) == (41, 25)
    np.testing.assert_array_equal(h.restarts, ho.restarts)
    np.testing.assert_allclose(np.diag(T), np.diag(To), rtol=1e-9)
    _, rel = _residuals(As, Q, T)
    _, rel_o = _residuals(As, Qo, To)
    assert rel.max() <= max(1.05 * rel_o.max(), 1e-13)

    # full size
    n = 1_508_065
    A = matrices.banded_csr(n, 35, 1234, planted=planted)
    assert A.nnz > 52_000_000
    np.random.seed(0)
    st = {}
    Q, T, h = amd.partial_schur(A, 20, stats=st, max_restarts=300)
    vals, rel = _residuals(A, Q, T)
    assert rel.max() < 5 * np.sqrt(np.finfo(np.float64).eps), rel
    assert np.all(np.abs(vals) > 25.0)              # the planted, dominant part of the spectrum
    np.testing.assert_allclose(Q.conj().T @ Q, np.eye(20), atol=1e-11)
    print(f"C3 stand-in: {st['restarts']} restarts, max rel residual {rel.max():.2e}")


def test_config3_shell_structured_stand_in(amd):
    """Config 3 once more with the STRUCTURE of the matrix it names (af_shell10: a shell finite-element model, 5 unknowns
    per node, 34.65 entries per row): matrices.shell_csr, 549 x 549 nodes = 1 507 005 rows, 52.6M entries in seven
    5 x 5 blocks per row a node row apart -- where the banded stand-in has one contiguous run."""
    from arnoldi_amd import matrices

    planted = tuple(60.0 - 1.5 * i for i in range(24))
    As = matrices.shell_csr(80, 75, 5, 1234, planted=planted)            # small twin against the oracle
    np.random.seed(0)
    Qo, To, ho = oracle.krylov_schur(As, 20)
    np.random.seed(0)
    st = {}
    Q, T, h = amd.partial_schur(As, 20, stats=st)
    assert (st["max_dim"], st["p"]) == (41, 25)
    np.testing.assert_array_equal(h.restarts, ho.restarts)
    np.testing.assert_allclose(np.diag(T), np.diag(To), rtol=1e-9)
    _, rel = _residuals(As, Q, T)
    _, rel_o = _residuals(As, Qo, To)
    assert rel.max() <= max(1.05 * rel_o.max(), 1e-13)

    A = matrices.shell_csr(549, 549, 5, 1234, planted=planted)           # full size
    assert A.shape[0] == 1_507_005 and A.nnz > 52_000_000
    np.random.seed(0)
    st = {}
    Q, T, h = amd.partial_schur(A, 20, stats=st, max_restarts=300)
    vals, rel = _residuals(A, Q, T)
    assert rel.max() < 5 * np.sqrt(np.finfo(np.float64).eps), rel
    assert np.all(np.abs(vals) > 25.0)              # the planted, dominant part of the spectrum
    np.testing.assert_allclose(Q.conj().T @ Q, np.eye(20), atol=1e-11)
    print(f"C3 shell stand-in: {st['restarts']} restarts, form {st['spmv_form']}, max rel residual {rel.max():.2e}")


def test_config4_laplace3d_16m_invariants(amd):
    import torch
    from arnoldi_amd import matrices
    from arnoldi_amd.krylov_schur import KrylovSchurSolver
    from arnoldi_amd.utils import arg_largest_magnitude

    dims = (251, 252, 253)
    A = matrices.laplace3d(*dims)
    n = A.shape[0]
    assert n == 16_002_756 and A.nnz == 111_638_270       # SURVEY 8 size table
    np.random.seed(0)
    s = KrylovSchurSolver(A, 10, 40, 15, 1e-8, arg_largest_magnitude)
    del A
    assert s.start() == 40
    for r in range(2):
        assert not s.contract(r)
        assert s.expand() == 40
    ctx, m = s.ctx, 40
    V = ctx.basis.V[:, :n]
    G = (V.conj() @ V.T).cpu().numpy()
    assert np.abs(G - np.eye(m + 1)).max() < 1e-11
    # A V[:, j] = V[:, :m+1] H[:, j]  for columns of the Krylov-Schur form (j < p: full column of H,
    # including the coupling row p) and of the Hessenberg tail
    Hd = torch.from_numpy(s.H).cuda()
    y = torch.empty(ctx.basis.ldv, dtype=torch.complex128, device="cuda")
    for j in (0, 7, 14, 15, 30, 39):
        ctx.op.apply(ctx.basis.col(j), y)
        r = y[:n] - (Hd[:, j].unsqueeze(0) @ V).squeeze(0)
        assert float(torch.linalg.norm(r)) < 1e-10, j
    # Ritz values are inside the analytic spectrum's hull [-12, 0]
    ritz = np.linalg.eigvals(s.H[:m, :m])
    assert ritz.real.min() > -12.0 and ritz.real.max() < 0.0 and np.abs(ritz.imag).max() < 1e-8


def test_config5_random_10m_planted_converges(amd):
    from arnoldi_amd import matrices

    n = 10_000_000
    planted = (4.0, 3.7, 3.4, 3.1, 2.8, 2.5)
    A = matrices.random_csr(n, 5, 1234, planted=planted)
    np.random.seed(0)
    st = {}
    Q, T, h = amd.partial_schur(A, 5, max_dim=20, stats=st)
    vals, rel = _residuals(A, Q, T)
    assert rel.max() < 5 * np.sqrt(np.finfo(np.float64).eps), rel
    # the planted diagonal entries dominate the bulk (radius ~ sqrt(5/3)); each eigenvalue sits
    # within O(bulk^2 / lambda) of its planted value
    np.testing.assert_allclose(np.sort(vals.real)[::-1], planted[:5], atol=0.2)
    # the same residuals evaluated on the device agree with the host's
    dvals, dres, drel = st["solver"].true_residuals()
    order_h, order_d = np.argsort(-vals.real), np.argsort(-dvals.real)
    np.testing.assert_allclose(dvals[order_d], vals[order_h], rtol=1e-12)
    np.testing.assert_allclose(drel[order_d], rel[order_h], rtol=1e-3, atol=1e-12)
    np.testing.assert_allclose(Q.conj().T @ Q, np.eye(5), atol=1e-11)
    assert st["restarts"] <= 30
    print(f"C5 planted: {st['restarts']} restarts, spmv form {st['solver'].op.spmv_form}, "
          f"max rel residual {rel.max():.2e}")
