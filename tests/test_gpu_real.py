"""GPU tests of the real-arithmetic mode: the real-vector SpMV kernels, the panel
kernels on real-packed columns (two rows per complex slot, reductions dropping the imaginary parts) and
``partial_schur(arithmetic="real")`` against the CPU oracle of the reference's complex iteration."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from conftest import torch_buffers
import real_cases as rc

pytestmark = pytest.mark.gpu

C128 = np.complex128


@pytest.fixture(scope="module")
def amd():
    from arnoldi_amd import mem

    assert mem.gpu_available(), "these tests need the MI355X"
    import arnoldi_amd
    from arnoldi_amd import _hip

    _hip.load()
    return arnoldi_amd


def _ragged(n, n_cols, seed):
    rng = np.random.default_rng(seed)
    lengths = rng.integers(0, 12, n)
    lengths[rng.integers(0, n, n // 10)] = 0
    if n > 20:
        lengths[5], lengths[17], lengths[18] = 700, 256, 257
    lengths = np.minimum(lengths, n_cols)
    lengths[0] = max(lengths[0], 1)
    indptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    cols = np.concatenate([np.sort(rng.choice(n_cols, L, replace=False)) for L in lengths if L > 0] or [np.zeros(0, int)])
    return sp.csr_matrix((rng.standard_normal(cols.size), cols.astype(np.int32), indptr), shape=(n, n_cols))


@pytest.mark.parametrize("shape", [(3, 5), (63, 63), (5000, 5000), (4001, 70001), (90001, 3000)])
@pytest.mark.parametrize("form", ["csr", "binned"])
def test_spmv_real_vectors(amd, shape, form):
    """y = A x and y += A x on float64 vectors, both SpMV forms, ragged rows, non-square blocks, odd
    sizes; bitwise reproducible."""
    torch = torch_buffers()
    from arnoldi_amd.device import DeviceCSR

    A = _ragged(shape[0], shape[1], shape[0] + shape[1])
    d = DeviceCSR(A)
    d.autotune(force=form)
    rng = np.random.default_rng(1)
    xh, y0 = rng.standard_normal(shape[1]), rng.standard_normal(shape[0])
    x = torch.from_numpy(xh).cuda()
    y = torch.from_numpy(y0.copy()).cuda()
    d.spmv(x, y, real=True)
    ref = A @ xh
    scale = np.abs(A).dot(np.abs(xh)).max() + 1e-300
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=1e-13 * scale)
    first = y.cpu().numpy().copy()
    d.spmv(x, y, real=True)
    np.testing.assert_array_equal(y.cpu().numpy(), first)                 # reproducible
    y.copy_(torch.from_numpy(y0))
    d.spmv(x, y, accumulate=True, real=True)
    np.testing.assert_allclose(y.cpu().numpy(), y0 + ref, rtol=0, atol=1e-13 * (scale + np.abs(y0).max()))


def test_spmv_real_matches_complex_kernel_on_packed_columns(amd):
    """The same product through the complex kernel (imaginary parts zero) and through the real kernel on a
    real-packed basis column."""
    from arnoldi_amd import device as dev, matrices

    A = matrices.random_csr(20001, 5, 3)
    d = dev.DeviceCSR(A)
    d.autotune(force="csr")
    xr = np.random.default_rng(0).standard_normal(20001)
    packed = dev.KrylovBasis(20001, 1, real=True)
    packed.set_col(0, xr)
    d.spmv(packed.col(0), packed.col(1), real=True)
    full = dev.KrylovBasis(20001, 1)
    full.set_col(0, xr.astype(C128))
    d.spmv(full.col(0), full.col(1))
    np.testing.assert_allclose(packed.get_cols(1, 2)[:, 0], full.get_cols(1, 2)[:, 0].real, rtol=0, atol=1e-13)
    assert packed.get_cols(0, 2).dtype == np.float64 and packed.n_rows == 10001 and packed.n_real == 20001


@pytest.mark.parametrize("n", [2, 63, 129, 1000, 100001])
def test_gather_f64(amd, n):
    torch = torch_buffers()
    from arnoldi_amd import device as dev

    rng = np.random.default_rng(n)
    src = rng.standard_normal(n)
    idx = rng.integers(0, n, 3 * n).astype(np.int32)
    s, i = torch.from_numpy(src).cuda(), torch.from_numpy(idx).cuda()
    out = torch.zeros(3 * n, dtype=torch.float64, device="cuda")
    dev.gather_f64(3 * n, i, s, out)
    np.testing.assert_array_equal(out.cpu().numpy(), src[idx])


@pytest.mark.parametrize("n,J", [(7, 2), (1001, 1), (1001, 7), (4096, 20), (30001, 33), (30001, 64)])
def test_packed_gram_schmidt_is_the_real_operation(amd, n, J):
    """dgks_gs on a real-packed panel == the oracle's dgks_gs on the real vectors: coefficients real, w
    and beta equal, second-pass decision equal (near-dependent w forces the second pass)."""
    torch = torch_buffers()
    from arnoldi_amd import device as dev

    rng = np.random.default_rng(n + J)
    V, _ = np.linalg.qr(rng.standard_normal((n, J)))
    for near in (False, True):
        w = rng.standard_normal(n)
        if near:
            w = V @ rng.standard_normal(J) + 1e-4 * w
        basis = dev.KrylovBasis(n, J, real=True)
        ws = dev.Workspace(basis.n_rows, J, real=True)
        basis.set_cols(0, V)
        basis.set_col(J, w)
        hdev = torch.zeros(J + 1, dtype=torch.complex128, device="cuda")
        dev.dgks_gs_device(basis, J, basis.col(J), hdev.data_ptr(), 1, 1e-8, ws, normalize=False)
        ctrl = ws.read_ctrl()
        wo, ho = w.astype(C128), np.zeros(J, C128)
        beta_o, broke_o, again_o = oracle.dgks_gs(wo, np.asfortranarray(V.astype(C128)), ho, 1e-8)
        h = hdev[:J].cpu().numpy()
        assert np.all(h.imag == 0.0)
        np.testing.assert_allclose(h.real, ho.real, rtol=0, atol=1e-12 * max(1.0, np.abs(ho).max()))
        np.testing.assert_allclose(basis.get_cols(J, J + 1)[:, 0], wo.real, rtol=0, atol=1e-12 * max(1.0, np.abs(w).max()))
        assert abs(ctrl.beta - beta_o) <= 1e-10 * max(beta_o, 1e-30) + 1e-14
        assert bool(ctrl.second_passes) == again_o == near and bool(ctrl.broken) == broke_o
        assert ctrl.real_mode == 1


def test_packed_truncate_and_arnoldi_invariants(amd):
    """Real-packed Arnoldi factorisation A V_m = V_{m+1} H (H real) and its compression by a real
    orthogonal Q, checked on the unpacked float64 columns."""
    from arnoldi_amd import matrices
    from arnoldi_amd.engine import ArnoldiContext, CsrOperator

    A = matrices.random_csr(50001, 5, 11)
    m, p = 24, 9
    ctx = ArnoldiContext(CsrOperator(A, real=True), m)
    rng = np.random.default_rng(0)
    v0 = rng.standard_normal(50001)
    ctx.set_start_vector(v0 / np.linalg.norm(v0))
    H = np.zeros((m + 1, m))
    assert ctx.expand(H, 0, m, 1e-8) == m
    V = ctx.basis.get_cols(0, m + 1)
    assert V.dtype == np.float64
    np.testing.assert_allclose(V.T @ V, np.eye(m + 1), atol=1e-12)
    np.testing.assert_allclose(A @ V[:, :m], V @ H, atol=1e-12 * np.abs(H).max())
    Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    ctx.truncate(Q[:, :p], m, p)
    W = ctx.basis.get_cols(0, p + 1)
    np.testing.assert_allclose(W[:, :p], V[:, :m] @ Q[:, :p], atol=1e-13)
    np.testing.assert_array_equal(W[:, p], V[:, m])


@pytest.mark.parametrize("name", sorted(["mark30_lr", "mark50_readme", "planted_odd_n", "laplace2d", "conjugate_pairs",
                                         "pair_cut_at_nev3", "pair_cut_at_nev5", "dense_array"]))
def test_real_arithmetic_solves(amd, name):
    A, nev, seed, kw = rc.cases()[name]
    rc.check_case(A, nev, seed, **kw)


def test_real_arithmetic_errors(amd):
    rc.check_errors()
    rc.check_auto()


def test_real_arithmetic_chained_path_is_bitwise_the_native_one(amd):
    """The Python-chained stage path (what several GPUs run) and the C-chained expansion give the same bits
    in real-packed mode too."""
    from arnoldi_amd import matrices
    from arnoldi_amd.engine import CsrOperator
    from arnoldi_amd.krylov_schur_real import RealKrylovSchurSolver

    A = matrices.random_csr(100_000, 5, 1234, planted=(4.0, 3.7, 3.4, 3.1, 2.8, 2.5))
    op = CsrOperator(A, real=True)
    out = []
    for chained in (False, True):
        np.random.seed(0)
        s = RealKrylovSchurSolver(op, 5, 20, 10, 1e-8, oracle.arg_largest_magnitude)
        s.ctx.force_chained = chained
        s.start()
        for r in range(50):
            if s.contract(r):
                break
            s.expand()
        out.append((s.result(), s.restarts_run))
    assert out[0][1] == out[1][1]
    np.testing.assert_array_equal(out[0][0][0], out[1][0][0])
    np.testing.assert_array_equal(out[0][0][1], out[1][0][1])


def test_real_arithmetic_full_size_planted(amd):
    """n = 4M planted random CSR: the real-packed solve finds the planted values; residuals on the host."""
    from arnoldi_amd import matrices

    n = 4_000_001
    planted = (4.0, 3.7, 3.4, 3.1, 2.8, 2.5)
    A = matrices.random_csr(n, 5, seed=1234, planted=planted)
    np.random.seed(0)
    st = {}
    Q, T, hist = amd.partial_schur(A, 5, max_dim=20, arithmetic="real", stats=st)
    np.testing.assert_allclose(np.sort(np.diag(T).real)[::-1], planted[:5], rtol=1e-6)
    assert np.abs(np.diag(T).imag).max() < 1e-9
    res = np.linalg.norm(A @ Q - Q @ T, axis=0)
    assert res.max() < 1e-6
    np.testing.assert_allclose(Q.conj().T @ Q, np.eye(5), atol=1e-10)


@pytest.mark.parametrize("which", ["LM", "LR"])
def test_real_arithmetic_stress_grid(amd, which):
    """The reference's stress grid (nev, ncv, p) in real arithmetic against the oracle of the complex
    iteration: on the symmetric Laplacian (real spectrum) the two iterations coincide restart for restart."""
    from arnoldi_amd import matrices
    from arnoldi_amd.harness import STRESS_GRID

    if which == "LM":
        A, sort_o, tol = matrices.laplace2d(30, 31), oracle.arg_largest_magnitude, None
    else:
        A, sort_o, tol = matrices.mark(44), oracle.arg_largest_real, 1e-8
    for nev, ncv, p in STRESS_GRID:
        kw = dict(max_dim=ncv, p=p, stopping_criterion=tol, max_restarts=4000, sort_function=sort_o)
        st = rc.check_case(A, nev, nev + ncv, **kw)
        if which == "LM":
            np.random.seed(nev + ncv)
            _, _, ho = oracle.krylov_schur(A, nev, **kw)
            assert int(st["restarts"]) == int(ho.restarts.max()), (nev, ncv, p)


@pytest.mark.parametrize("n", [3, 4, 5, 17, 63, 64, 65, 129, 257])
def test_real_arithmetic_tiny_sizes(amd, n):
    """Odd and tiny dimensions (panels of 2 .. 129 packed rows, Krylov space = whole space for n <= 20):
    the dominant eigenpair of a dense real matrix, against the complex path and numpy."""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)) + np.diag(np.arange(n) * 2.0)
    m = min(20, n)
    out = {}
    for arith in ("real", "complex"):
        np.random.seed(1)
        Q, T, _ = amd.partial_schur(A, 1, max_dim=m, arithmetic=arith, max_restarts=500)
        assert Q.shape == (n, 1)
        out[arith] = T[0, 0]
        assert np.linalg.norm(A @ Q - Q @ T) < 1e-5 * abs(T[0, 0])
    ev = np.linalg.eigvals(A)
    fold = lambda z: complex(z.real, abs(z.imag))          # noqa: E731  (either member of a dominant pair)
    assert abs(fold(out["real"]) - fold(out["complex"])) < 1e-7 * abs(out["complex"])
    assert np.abs(ev - out["real"]).min() < 1e-6 * abs(out["real"])


# ---- the real-arithmetic counterparts of the widened solvers (no more "complex only") ---------------------------------
@pytest.mark.parametrize("name", ["mark50_readme", "laplace2d", "pair_cut_at_nev5", "conjugate_pairs"])
def test_real_arithmetic_locking(amd, name):
    """partial_schur(arithmetic="real", locking=True): eigenvalues of the oracle to 10 tol, residuals <= 1.05 x the
    oracle's (floor 10 tol), conjugate pairs locked as a whole."""
    A, nev, seed, kw = rc.cases()[name]
    rc.check_locking_real(A, nev, seed, **{k: v for k, v in kw.items() if k != "max_restarts"})


def test_real_arithmetic_deflate_and_residual_norms(amd, monkeypatch):
    rc.check_deflate_real()
    rc.check_residual_norms_real()
    # the same with every operator in the binned form: the expansions defer their normalisations, so the breakdown
    # happens among raw columns and the deflating compression folds their scales (complex and real drivers)
    monkeypatch.setenv("AKS_SPMV_FORM", "binned")
    monkeypatch.setenv("AKS_DEFER_MAX_STEPS", "40")         # (the cases expand by 20 steps; the default defers up to 12)
    rc.check_deflate_real()
    rc.check_residual_norms_real()


@pytest.mark.parametrize("name", ["mark30_lr", "pair_cut_at_nev3", "planted_odd_n"])
def test_real_arithmetic_explicit_restarts_with_deflation(amd, name):
    """explicit_restarts_with_deflation(arithmetic="real") (explicit_restarts.py:80-168 in real arithmetic): real
    Ritz values as in the reference, conjugate pairs as two real Schur vectors; against the oracle of the
    reference's complex solver."""
    A, nev, seed, kw = rc.cases()[name]
    kw = {k: v for k, v in kw.items() if k != "max_restarts"}
    rc.check_explicit_deflation_real(A, min(nev, 3), seed, max_restarts=400, **kw)
