#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Build-container only: imports ``arnoldi`` from ``/root/reference/src`` (which
does not exist on the GPU box) and writes small ``.npz`` files holding inputs
and the reference's outputs.  Nothing of the reference's source is stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

REF = "/root/reference/src"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from arnoldi.decomposition import RitzDecomposition, arnoldi_decomposition  # noqa: E402
from arnoldi.explicit_restarts import (  # noqa: E402
    explicit_restarts_with_deflation,
    mgs,
    naive_explicit_restarts,
)
from arnoldi.krylov_schur import partial_schur  # noqa: E402
from arnoldi.matrices import laplace, laplace_eigen, mark  # noqa: E402
from arnoldi.ortho import dgks_gs  # noqa: E402
from arnoldi.utils import (  # noqa: E402
    arg_largest_magnitude,
    arg_largest_real,
    ordered_schur,
    rand_normalized_vector,
)

HERE = os.path.dirname(os.path.abspath(__file__))
C128 = np.complex128


def csr_parts(A, prefix):
    A = sp.csr_matrix(A)
    A.sum_duplicates()
    A.sort_indices()
    return {
        prefix + "_indptr": A.indptr.astype(np.int32),
        prefix + "_indices": A.indices.astype(np.int32),
        prefix + "_data": A.data,
        prefix + "_shape": np.array(A.shape, np.int64),
    }


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def residuals(A, Q, T):
    vals, S = np.linalg.eig(T)
    vecs = Q @ S
    r = np.linalg.norm(A @ vecs - vecs * vals, axis=0) / np.abs(vals)
    return vals, r


def solve_record(A, seed, **kw):
    np.random.seed(seed)
    v0 = rand_normalized_vector(A.shape[0], C128)
    np.random.seed(seed)
    Q, T, hist = partial_schur(A, **kw)
    vals, r = residuals(A, Q, T)
    return {
        "v0": v0,
        "Q": np.array(Q),
        "T": np.array(T),
        "hist_matvecs": hist.matvecs,
        "hist_restarts": hist.restarts,
        "eigvals": vals,
        "rel_residuals": r,
    }


def laplace2d(nx, ny):
    """5-point stencil (-4 / +1) on an nx x ny grid: kron of the reference's laplace()."""
    Lx = sp.csr_matrix(laplace(nx))
    Ly = sp.csr_matrix(laplace(ny))
    return (sp.kron(sp.eye(ny), Lx) + sp.kron(Ly, sp.eye(nx))).tocsr()


def random_csr(n, per_row, seed, planted=None):
    rng = np.random.default_rng(seed)
    idx = np.sort(rng.integers(0, n, (n, per_row), dtype=np.int64), axis=1).astype(np.int32)
    data = rng.uniform(-1.0, 1.0, (n, per_row))
    indptr = np.arange(0, per_row * n + 1, per_row, dtype=np.int32)
    A = sp.csr_matrix((data.ravel(), idx.ravel(), indptr), shape=(n, n))
    A.sum_duplicates()
    if planted is not None:
        rows = rng.choice(n, size=len(planted), replace=False)
        A = A.tolil()
        for r, val in zip(rows, planted):
            A[r, r] = val
        A = A.tocsr()
    A.sort_indices()
    return A


def main():
    # ---- G1: literal matrices -------------------------------------------------
    g1 = {}
    for m in (2, 3, 10, 50):
        g1.update(csr_parts(mark(m), f"mark{m}"))
    g1.update(csr_parts(laplace(5), "laplace5"))
    g1["laplace_eigen5"] = laplace_eigen(5)
    g1["laplace_eigen100"] = laplace_eigen(100)
    save("g1_matrices", **g1)

    # ---- G5: dgks_gs, with and without the second pass -------------------------
    rng = np.random.default_rng(7)
    n, J = 400, 9
    Vq, _ = np.linalg.qr(rng.standard_normal((n, J)) + 1j * rng.standard_normal((n, J)))
    V = np.asfortranarray(Vq.astype(C128))
    g5 = {"V": V}
    # (a) generic vector: single pass; (b) nearly in span(V): second pass; (c) in span: breakdown
    generic = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(C128)
    near = (V @ (rng.standard_normal(J) + 1j * rng.standard_normal(J)) + 1e-3 * generic).astype(C128)
    inside = (V @ (rng.standard_normal(J) + 1j * rng.standard_normal(J))).astype(C128)
    for tag, w0 in (("generic", generic), ("near", near), ("inside", inside)):
        w = w0.copy()
        h = np.zeros(J, C128)
        beta, broke = dgks_gs(w, V, h, 1e-8)
        g5[f"{tag}_w_in"] = w0
        g5[f"{tag}_w_out"] = w
        g5[f"{tag}_h"] = h
        g5[f"{tag}_beta"] = np.float64(beta)
        g5[f"{tag}_breakdown"] = np.bool_(broke)
    save("g5_dgks_gs", **g5)

    # ---- G4: Arnoldi expansions -------------------------------------------------
    g4 = {}
    A = mark(10)
    np.random.seed(3)
    v0 = rand_normalized_vector(A.shape[0], C128)
    m = 6
    V = np.zeros((A.shape[0], m + 1), C128, order="F")
    H = np.zeros((m + 1, m), C128)
    V[:, 0] = v0
    _, _, n_iter = arnoldi_decomposition(A, V, H, 1e-8)
    g4.update(csr_parts(A, "mark10"))
    g4.update({"mark10_v0": v0, "mark10_V": V, "mark10_H": H, "mark10_niter": np.int64(n_iter)})
    # resume from start_dim (the restart seam): first 3 columns, then 3..6
    V2 = np.zeros_like(V)
    H2 = np.zeros_like(H)
    V2[:, 0] = v0
    arnoldi_decomposition(A, V2, H2, 1e-8, max_dim=3)
    g4.update({"mark10_V_first3": V2.copy(), "mark10_H_first3": H2.copy()})
    arnoldi_decomposition(A, V2, H2, 1e-8, start_dim=3, max_dim=m)
    g4.update({"mark10_V_resumed": V2, "mark10_H_resumed": H2})

    # complex sparse + I, C-ordered V (tests/test_decomposition.py:71-90 shape)
    n = 10
    rs = np.random.RandomState(11)
    Ac = sp.random(n, n, density=5 / n, dtype=C128, random_state=rs) + sp.diags_array(np.ones(n))
    Ac = sp.csr_matrix(Ac)
    np.random.seed(4)
    v0c = rand_normalized_vector(n, C128)
    Vc = np.zeros((n, m + 1), C128)
    Hc = np.zeros((m + 1, m), C128)
    Vc[:, 0] = v0c
    _, _, n_iter_c = arnoldi_decomposition(Ac, Vc, Hc, 1e-8)
    g4.update(csr_parts(Ac, "cplx"))
    g4.update({"cplx_v0": v0c, "cplx_V": Vc, "cplx_H": Hc, "cplx_niter": np.int64(n_iter_c)})

    # breakdown: v0 is an eigenvector => one iteration (tests/test_decomposition.py:115-139)
    evals, evecs = np.linalg.eig(Ac.toarray())
    vb = evecs[:, np.argmax(np.abs(evals))].astype(C128)
    Vb = np.zeros((n, m + 1), C128, order="F")
    Hb = np.zeros((m + 1, m), C128)
    Vb[:, 0] = vb
    Vv, Hv, n_iter_b = arnoldi_decomposition(Ac, Vb, Hb, 1e-8)
    g4.update({"brk_v0": vb, "brk_V": Vb, "brk_H": Hb, "brk_niter": np.int64(n_iter_b),
               "brk_Vshape": np.array(Vv.shape), "brk_Hshape": np.array(Hv.shape)})
    save("g4_arnoldi", **g4)

    # ---- G6: ordered_schur -------------------------------------------------------
    r_T = np.array([
        [5.0, 1.5, 0.8, 0.1, 0.4],
        [0.0, 4.0, 1.2, 1.0, 0.5],
        [0.0, 0.0, 3.0, 1.0, 0.3],
        [0.0, 0.0, 0.0, 2.0, 0.6],
        [0.0, 0.0, 0.0, 0.0, 1.0],
    ])
    rs = np.random.RandomState(5)
    g6 = {}
    for ch in ("F", "D"):
        rq, _ = np.linalg.qr(rs.randn(5, 5).astype(ch))
        a = rq.T @ r_T.astype(ch) @ rq
        T, Z = ordered_schur(a, output="complex", sort_function=lambda v: np.argsort(v))
        g6[f"{ch}_a"] = a
        g6[f"{ch}_T"] = T
        g6[f"{ch}_Z"] = Z
    # a complex Hessenberg-like input with both sort keys
    hm = (rs.randn(12, 12) + 1j * rs.randn(12, 12)).astype(C128)
    hm = np.triu(hm, -1)
    for tag, fn in (("lm", arg_largest_magnitude), ("lr", arg_largest_real)):
        T, Z = ordered_schur(hm, output="complex", sort_function=fn)
        g6[f"hess_{tag}_T"] = T
        g6[f"hess_{tag}_Z"] = Z
    g6["hess_a"] = hm
    save("g6_ordered_schur", **g6)

    # ---- G2/G3: Markov solves ---------------------------------------------------
    g3 = {}
    A10 = mark(10)
    rec = solve_record(A10, 0, nev=3, max_dim=5, sort_function=arg_largest_real, max_restarts=1000)
    g3.update({f"mark10_s0_{k}": v for k, v in rec.items()})
    A50 = mark(50)
    for seed in (0, 1):
        rec = solve_record(A50, seed, nev=5, max_dim=20, stopping_criterion=1e-8,
                           sort_function=arg_largest_real)
        g3.update({f"mark50_s{seed}_{k}": v for k, v in rec.items()})
    # default-arguments path (tol = sqrt(eps), p, max_dim defaults), LR
    rec = solve_record(A50, 2, nev=4, sort_function=arg_largest_real)
    g3.update({f"mark50_defaults_{k}": v for k, v in rec.items()})
    save("g3_markov", **g3)

    # ---- dense operator (tests/test_krylov_schur.py:28-49) -----------------------
    rs = np.random.RandomState(9)
    D = np.diag([7, 7, 5, 4, 3, 2, 1]).astype(float)
    qq, _ = np.linalg.qr(rs.randn(7, 7))
    Ad = qq.T @ D @ qq
    rec = solve_record(Ad, 0, nev=3, max_dim=6, sort_function=arg_largest_real, max_restarts=1000)
    gd = {f"diag_{k}": v for k, v in rec.items()}
    gd["diag_A"] = Ad
    save("g2_dense_diag", **gd)

    # ---- G7: 2-D Laplace 30x31, LM ------------------------------------------------
    L = laplace2d(30, 31)
    rec = solve_record(L, 0, nev=10, max_dim=40, sort_function=arg_largest_magnitude)
    g7 = {f"lap_{k}": v for k, v in rec.items() if k != "Q"}
    g7.update(csr_parts(L, "lap"))
    ex = laplace_eigen(30)[:, None] + laplace_eigen(31)[None, :]
    g7["lap_analytic"] = np.sort(ex.ravel())
    save("g7_laplace2d", **g7)

    # ---- G8: planted-spectrum random CSR (config-5 shape, small n) ----------------
    n = 20000
    Ar = random_csr(n, 5, 1234, planted=(4.0, 3.7, 3.4, 3.1, 2.8, 2.5))
    g8 = {"n": np.int64(n)}
    for seed in (0, 1):
        rec = solve_record(Ar, seed, nev=5, max_dim=20, sort_function=arg_largest_magnitude)
        g8.update({f"s{seed}_{k}": v for k, v in rec.items() if k not in ("Q", "v0")})
    save("g8_random_planted", **g8)

    # ---- non-convergence: message + count (krylov_schur.py:108-109) ---------------
    np.random.seed(0)
    try:
        partial_schur(random_csr(2000, 5, 1234), 5, max_dim=20, max_restarts=3)
        msg = ""
    except ValueError as e:
        msg = str(e)
    save("g9_errors", not_converged=np.array(msg))


def explicit_restart_fixtures():
    """G10: the explicit-restart path: RitzDecomposition, mgs and the two
    solvers of src/arnoldi/explicit_restarts.py on the cases of tests/test_explicit_restarts.py
    plus two sparse ones.  ``python make_golden.py explicit`` regenerates only this file."""
    g = {}
    rng = np.random.default_rng(11)

    # (a) Ritz extraction from a real Arnoldi factorisation of mark(10), m = 8
    A = mark(10)
    n, m = A.shape[0], 8
    V = np.zeros((n, m + 1), C128)
    H = np.zeros((m + 1, m), C128)
    np.random.seed(3)
    V[:, 0] = rand_normalized_vector(n, C128)
    Va, Ha, n_iter = arnoldi_decomposition(A, V, H)
    assert n_iter == m
    g["ritz_V"], g["ritz_H"] = np.array(Va), np.array(Ha)
    for tag, nr, fn in (("lm3", 3, None), ("lr8", 8, arg_largest_real)):
        r = RitzDecomposition.from_v_and_h(Va, Ha, nr, sort_function=fn)
        g[f"ritz_{tag}_values"] = r.values
        g[f"ritz_{tag}_vectors"] = r.vectors
        g[f"ritz_{tag}_approx"] = r.approximate_residuals
        g[f"ritz_{tag}_true"] = r.compute_true_residuals(A)

    # (b) mgs against 0, 1 and 6 orthonormal columns
    nn = 300
    Bq, _ = np.linalg.qr(rng.standard_normal((nn, 6)) + 1j * rng.standard_normal((nn, 6)))
    g["mgs_basis"] = np.asfortranarray(Bq.astype(C128))
    w0 = (rng.standard_normal(nn) + 1j * rng.standard_normal(nn)).astype(C128)
    g["mgs_w_in"] = w0
    for k in (0, 1, 6):
        w = w0.copy()
        mgs(g["mgs_basis"][:, :k], w, 1e-8)
        g[f"mgs_w_out_{k}"] = w

    # (c) naive explicit restarts: Saad table 6.2 (mark(10), m = 10) and the convergence test
    for restarts in (1, 2, 3, 4, 5):
        np.random.seed(0)
        ritz, ok, used = naive_explicit_restarts(A, 10, max_restarts=restarts)
        g[f"naive_r{restarts}_value"] = ritz.values
        g[f"naive_r{restarts}_true"] = ritz.compute_true_residuals(A)
        g[f"naive_r{restarts}_flags"] = np.array([int(ok), used])
    np.random.seed(0)
    ritz, ok, used = naive_explicit_restarts(A, 20, max_restarts=200, stopping_criterion=1e-6)
    g["naive_conv_value"], g["naive_conv_true"] = ritz.values, ritz.compute_true_residuals(A)
    g["naive_conv_flags"] = np.array([int(ok), used])
    g["naive_conv_vector"] = ritz.vectors[:, 0]

    # (d) deflation solver: Saad table 6.3 (mark(10), m = 10, k = 3, LR), the rotated diagonal with a
    # double eigenvalue (happy breakdown inside), mark(30) LR and a 2-D Laplacian LM
    def record(tag, M, nev, seed, **kw):
        np.random.seed(seed)
        vals, vecs, hist = explicit_restarts_with_deflation(M, nev, **kw)
        g[f"{tag}_vals"], g[f"{tag}_vecs"] = vals, vecs
        g[f"{tag}_matvecs"], g[f"{tag}_restarts"] = hist.matvecs, hist.restarts
        g[f"{tag}_residuals"] = np.linalg.norm(M @ vecs - vals * vecs, axis=0)

    record("defl_mark10", A, 3, 0, max_dim=10, stopping_criterion=1e-8, sort_function=arg_largest_real)
    D = np.diag([7.0, 7, 5, 4, 3, 2, 1])
    Qr, _ = np.linalg.qr(rng.standard_normal((7, 7)))
    Ad = Qr.T @ D @ Qr
    g["defl_diag_A"] = Ad
    record("defl_diag", Ad, 3, 0)
    record("defl_mark30", mark(30), 4, 1, max_dim=30, stopping_criterion=1e-8, sort_function=arg_largest_real)
    L = laplace2d(12, 13)
    record("defl_lap", L, 3, 2, max_dim=30, stopping_criterion=1e-6, max_restarts=400)

    # (e) the error of test_fail_convergence
    np.random.seed(0)
    try:
        explicit_restarts_with_deflation(A, 3, max_dim=5, stopping_criterion=1e-16, max_restarts=10)
        msg = ""
    except ValueError as e:
        msg = str(e)
    g["defl_fail_message"] = np.array(msg)
    save("g10_explicit_restarts", **g)


if __name__ == "__main__":
    if sys.argv[1:] == ["explicit"]:
        explicit_restart_fixtures()
    else:
        main()
        explicit_restart_fixtures()
