"""Test / benchmark matrix generators (host side, vectorised).

``mark``, ``laplace`` and ``laplace_eigen`` reproduce src/arnoldi/matrices.py:5-95
entry for entry; ``laplace2d``, ``laplace3d`` and ``random_csr`` build the larger
BASELINE.json configurations, which the reference has no generator for.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def mark(m):
    """Markov chain of a random walk on an m-row triangular grid (Saad, section 2.5.1).

    n = m(m+1)/2 states numbered row by row; from state (i, j) the walk moves "up"
    (to j+1 or i+1) with probability ``pd = (i+j+1)/(2(m-1))`` each and "down" (to j-1 or
    i-1) with ``pu = 1/2 - (i+j-1)/(2(m-1))`` each; moves that would leave the grid on
    the i == 0 / j == 0 edges are folded back, which doubles that entry.  The matrix is
    built from whole-grid index arrays (no Python loop over states).
    """
    n = m * (m + 1) // 2
    cst = 0.5 / (m - 1)
    i = np.repeat(np.arange(m), np.arange(m, 0, -1))          # grid row of every state
    first = np.cumsum(np.concatenate([[0], np.arange(m, 1, -1)]))  # state id of (i, 0)
    ix = np.arange(n)
    j = ix - first[i]
    width = m - i                                              # states in grid row i
    pd = cst * (i + j + 1)
    pu = 0.5 - cst * (i + j - 1)

    up = j < width - 1
    north = (ix[up], ix[up] + 1, pd[up] * np.where(i[up] == 0, 2.0, 1.0))
    east = (ix[up], ix[up] + width[up], pd[up] * np.where(j[up] == 0, 2.0, 1.0))
    s = j > 0
    south = (ix[s], ix[s] - 1, pu[s])
    wst = i > 0
    west = (ix[wst], ix[wst] - width[wst] - 1, pu[wst])

    rows = np.concatenate([north[0], east[0], south[0], west[0]])
    cols = np.concatenate([north[1], east[1], south[1], west[1]])
    vals = np.concatenate([north[2], east[2], south[2], west[2]])
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    A.sort_indices()
    return A


def laplace_eigen(n):
    """Eigenvalues of ``laplace(n)``: -2 + 2 cos(k pi / (n+1)), k = 1..n."""
    k = np.arange(1, n + 1)
    return -2 + 2 * np.cos(k * np.pi / (n + 1))


def laplace(n, dtype=None):
    """1-D discrete Laplacian: tridiagonal with -2 on the diagonal and 1 beside it."""
    ones = np.ones(n - 1, dtype=dtype)
    return sp.diags_array([-2 * np.ones(n, dtype=dtype), ones, ones], offsets=[0, -1, 1])


def laplace2d(nx, ny, dtype=np.float64):
    """5-point Laplacian (-4 / +1) on an nx x ny grid, x fastest; n = nx*ny.
    Eigenvalues: laplace_eigen(nx)[a] + laplace_eigen(ny)[b]."""
    return _stencil((nx, ny), dtype)


def laplace3d(nx, ny, nz, dtype=np.float64):
    """7-point Laplacian (-6 / +1) on an nx x ny x nz grid, x fastest."""
    return _stencil((nx, ny, nz), dtype)


def _stencil(dims, dtype, row_range=None):
    """CSR rows ``row_range`` (default all) of the (2d+1)-point Laplacian on a grid."""
    n = int(np.prod(dims))
    r0, r1 = (0, n) if row_range is None else row_range
    rows = np.arange(r0, r1, dtype=np.int64)
    strides = np.concatenate([[1], np.cumprod(dims[:-1])]).astype(np.int64)
    coord = [(rows // s) % d for s, d in zip(strides, dims)]
    # neighbours in increasing column order: -s_k (k = d-1..0), 0, +s_k (k = 0..d-1)
    cand_cols, cand_ok = [], []
    for k in reversed(range(len(dims))):
        cand_cols.append(rows - strides[k]); cand_ok.append(coord[k] > 0)
    cand_cols.append(rows); cand_ok.append(np.ones_like(rows, dtype=bool))
    for k in range(len(dims)):
        cand_cols.append(rows + strides[k]); cand_ok.append(coord[k] < dims[k] - 1)
    ok = np.stack(cand_ok, axis=1)
    cols = np.stack(cand_cols, axis=1)
    vals = np.ones(ok.shape, dtype=dtype)
    vals[:, len(dims)] = -2.0 * len(dims)
    indptr = np.concatenate([[0], np.cumsum(ok.sum(axis=1))])
    idx_dtype = np.int32 if n < 2**31 - 1 and indptr[-1] < 2**31 - 1 else np.int64
    return sp.csr_matrix((vals[ok], cols[ok].astype(idx_dtype), indptr.astype(idx_dtype)),
                         shape=(r1 - r0, n))


def laplace_rows(dims, r0, r1, dtype=np.float64):
    """Rows r0..r1 (global column ids) of the grid Laplacian -- what one rank of a
    row-sharded solve builds without materialising the whole matrix."""
    return _stencil(tuple(dims), dtype, (r0, r1))


def random_csr(n, per_row=5, seed=1234, planted=None, row_range=None):
    """BASELINE config 5: ``per_row`` uniformly random columns per row (sorted, duplicates
    summed), values U(-1, 1), ``default_rng(seed)``.

    ``planted``: optional diagonal values written on random rows -- gives the otherwise
    hard spectrum a few dominant eigenvalues so that the solve converges.
    ``row_range``: build only rows r0..r1 (same numbers as the full matrix: the generator
    is advanced block by block in a fixed order).
    """
    rng = np.random.default_rng(seed)
    r0, r1 = (0, n) if row_range is None else row_range
    block = 1 << 20
    cols_parts, vals_parts = [], []
    for b0 in range(0, n, block):
        b1 = min(n, b0 + block)
        c = rng.integers(0, n, (b1 - b0, per_row), dtype=np.int64)
        v = rng.uniform(-1.0, 1.0, (b1 - b0, per_row))
        lo, hi = max(b0, r0), min(b1, r1)
        if lo < hi:
            cols_parts.append(c[lo - b0: hi - b0])
            vals_parts.append(v[lo - b0: hi - b0])
    cols = np.concatenate(cols_parts)
    vals = np.concatenate(vals_parts)
    order = np.argsort(cols, axis=1, kind="stable")
    cols = np.take_along_axis(cols, order, axis=1)
    vals = np.take_along_axis(vals, order, axis=1)
    nloc = r1 - r0
    indptr = np.arange(0, per_row * nloc + 1, per_row, dtype=np.int64)
    A = sp.csr_matrix((vals.ravel(), cols.ravel(), indptr), shape=(nloc, n))
    A.sum_duplicates()
    if planted is not None:
        prng = np.random.default_rng(seed + 1)
        where = prng.choice(n, size=len(planted), replace=False)
        add_r, add_v = [], []
        for r, val in zip(where, planted):
            if r0 <= r < r1:
                add_r.append(r - r0)
                add_v.append(val - A[r - r0, r])
        if add_r:
            D = sp.csr_matrix((add_v, (add_r, np.asarray(add_r) + r0)), shape=A.shape)
            A = (A + D).tocsr()
    A.sort_indices()
    if A.nnz < 2**31 - 1 and n < 2**31 - 1:
        A = sp.csr_matrix((A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32)), shape=A.shape)
    return A


def banded_csr(n, per_row=35, seed=1234, planted=None, dtype=np.float64):
    """Stand-in for BASELINE config 3 (SuiteSparse af_shell10 is not available offline): a
    symmetric-pattern band of ``per_row`` entries per row (offsets -h..h, h = per_row // 2,
    clipped at the ends), off-diagonal values U(-1, 1), diagonal U(0, 1) * per_row / 4.
    ``planted``: diagonal values written on evenly spaced rows (dominant eigenvalues)."""
    rng = np.random.default_rng(seed)
    h = per_row // 2
    offs = np.arange(-h, h + 1, dtype=np.int64)
    rows = np.arange(n, dtype=np.int64)
    cols = rows[:, None] + offs[None, :]
    ok = (cols >= 0) & (cols < n)
    vals = rng.uniform(-1.0, 1.0, cols.shape).astype(dtype)
    vals[:, h] = rng.uniform(0.0, 1.0, n) * per_row / 4
    if planted is not None:
        where = np.linspace(n // 7, n - n // 7, len(planted)).astype(np.int64)
        vals[where, h] = planted
    indptr = np.concatenate([[0], np.cumsum(ok.sum(axis=1))])
    return sp.csr_matrix((vals[ok], cols[ok].astype(np.int32), indptr.astype(np.int32)), shape=(n, n))


def shell_csr(nx, ny, dof=5, seed=1234, planted=None, dtype=np.float64):
    """A second stand-in for BASELINE config 3 with the STRUCTURE of a shell finite-element matrix such as af_shell10
    (n = 1 508 065, 34.65 entries per row; not available offline): ``dof`` unknowns per node of an ``nx x ny``
    triangulated sheet, every node coupled to itself and its six mesh neighbours (i +- 1, j), (i, j +- 1), (i + 1, j - 1),
    (i - 1, j + 1) by dense ``dof x dof`` blocks -- 7 * dof = 35 entries per interior row in seven runs of ``dof``
    consecutive columns, a node row apart, instead of ``banded_csr``'s one contiguous run.  Symmetric pattern,
    non-symmetric values: off-diagonal U(-1, 1), diagonal U(0, 1) * 7 * dof / 4.  ``planted``: diagonal values written on
    evenly spaced rows (dominant eigenvalues).  ``nx = ny = 549`` gives n = 1 507 005, nnz = 52.6M."""
    rng = np.random.default_rng(seed)
    nodes = nx * ny
    i, j = np.arange(nodes, dtype=np.int64) % nx, np.arange(nodes, dtype=np.int64) // nx
    steps = ((0, -1), (1, -1), (-1, 0), (0, 0), (1, 0), (-1, 1), (0, 1))              # ascending node number
    nbr = np.empty((nodes, 7), dtype=np.int64)
    ok = np.empty((nodes, 7), dtype=bool)
    for k, (di, dj) in enumerate(steps):
        ii, jj = i + di, j + dj
        ok[:, k] = (ii >= 0) & (ii < nx) & (jj >= 0) & (jj < ny)
        nbr[:, k] = jj * nx + ii
    n = nodes * dof
    # row (node, d) holds, per valid neighbour, the columns dof * nbr + (0 .. dof-1)
    cols = np.empty((nodes, dof, 7, dof), dtype=np.int32)
    cols[:] = ((dof * nbr).astype(np.int32)[:, :, None] + np.arange(dof, dtype=np.int32))[:, None, :, :]
    cols = cols.reshape(n, 7 * dof)
    keep = np.empty((nodes, dof, 7, dof), dtype=bool)
    keep[:] = ok[:, None, :, None]
    keep = keep.reshape(n, 7 * dof)
    vals = rng.random((n, 7 * dof))                                                    # U(-1, 1), in place
    vals *= 2.0
    vals -= 1.0
    vals = vals.astype(dtype, copy=False)
    diag_slot = 3 * dof + (np.arange(n, dtype=np.int64) % dof)                          # the row's own column
    rows = np.arange(n, dtype=np.int64)
    vals[rows, diag_slot] = rng.uniform(0.0, 1.0, n) * 7 * dof / 4
    if planted is not None:
        where = np.linspace(n // 7, n - n // 7, len(planted)).astype(np.int64)
        vals[where, diag_slot[where]] = planted
    indptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))])
    return sp.csr_matrix((vals[keep], cols[keep], indptr.astype(np.int32)), shape=(n, n))

