"""Benchmark / comparison harness around ``partial_schur``.

Counterpart of how the reference is timed and checked by its own scripts
(scripts/utils.py:29-187, scripts/benchmark-partial-schur.py, scripts/stress-test.py):

  * ``EigensolverParameters`` / ``Statistics``     run description and outcome
  * ``krylov_schur_eig``     eigenpairs from the partial Schur form, sorted by ``which``
  * ``arpack_eig``           SciPy ARPACK (``eigs``) as the independent comparator, with a
                             matvec counter
  * ``find_best_matching``   Hungarian matching of two eigenvalue sets
  * ``true_residuals`` / ``residual_report``   ||A v - l v|| and ||A v - l v|| / |l|
  * ``compare`` / ``sweep``  one comparison row / a CSV over a parameter grid
  * ``load_matrix``          SuiteSparse MATLAB ``.mat`` (``Problem.A``), MatrixMarket ``.mtx`` or
                             SciPy ``.npz`` into canonical CSR

SLEPc / PETSc comparators of the reference are not reproduced (not installable offline).
Everything here is host-side orchestration; the solve itself runs on the GPU.
"""
from __future__ import annotations

import csv
import dataclasses
import os
import time

import numpy as np
import scipy.io
import scipy.sparse as sp
from scipy.optimize import linear_sum_assignment
from scipy.sparse.linalg import LinearOperator, eigs

from .krylov_schur import partial_schur
from .utils import arg_largest_magnitude, arg_largest_real

WHICH_TO_SORT = {"LM": arg_largest_magnitude, "LR": arg_largest_real}

# (nev, ncv, p) grid of the reference's stress test (scripts/stress-test.py:29-41)
STRESS_GRID = [(3, 20, 10), (6, 20, 12), (10, 20, 16), (12, 30, 21), (20, 40, 30), (30, 50, 40), (50, 80, 65),
               (50, 100, 75), (75, 100, 85)]


@dataclasses.dataclass
class EigensolverParameters:
    nev: int = 6
    ncv: int = 20
    tol: float = 1e-8
    max_restarts: int = 1000
    p: int | None = None
    which: str = "LM"


@dataclasses.dataclass
class Statistics:
    elapsed: float = 0.0
    dtype: np.dtype = dataclasses.field(default_factory=lambda: np.dtype("complex128"))
    matvecs: int = 0          # operator applications actually performed
    restarts: int = 0
    booked_matvecs: int = 0   # max(History.matvecs): the reference's own (over-)estimate


class MatvecCounter(LinearOperator):
    """Counts operator applications (used for the ARPACK side; the device solver reports its
    own count, so the matrix itself can stay resident in HBM)."""

    def __init__(self, A):
        self.A = A
        self.matvecs = 0
        super().__init__(dtype=np.dtype(A.dtype), shape=A.shape)

    def _matvec(self, x):
        self.matvecs += 1
        return self.A @ x

    def _rmatvec(self, x):
        self.matvecs += 1
        return self.A.conj().T @ x


def find_best_matching(a, b):
    """Pair the entries of two equally long eigenvalue arrays so that the total distance is
    minimal (Hungarian algorithm); returns the two arrays in matched order."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"Shape mismatch: {a.shape} vs {b.shape}"
    rows, cols = linear_sum_assignment(np.abs(a[:, None] - b[None, :]))
    return a[rows], b[cols]


def true_residuals(A, vals, vecs):
    """(||A v_k - l_k v_k||, the same divided by |l_k|) for every pair."""
    res = np.linalg.norm(A @ vecs - vecs * vals, axis=0)
    return res, res / np.abs(vals)


def residual_report(label, A, vals, vecs):
    res, rel = true_residuals(A, vals, vecs)
    lines = [f"--- True residuals: {label} ---"]
    for k, (val, r, q) in enumerate(zip(vals, res, rel)):
        lines.append(f"  eigval[{k}] = {val.real:+.6g}{val.imag:+.6g}j    |Av-lv|={r:.3e}    |Av-lv|/|l|={q:.3e}")
    return "\n".join(lines)


def krylov_schur_eig(A, parameters: EigensolverParameters, **solver_kw):
    """Run the device solver; eigenpairs ``vals, S = eig(T); vecs = Q @ S`` sorted by ``which``
    (README.md:47-48 of the reference)."""
    stats = {}
    t0 = time.perf_counter()
    Q, T, history = partial_schur(
        A, parameters.nev, max_dim=parameters.ncv, stopping_criterion=parameters.tol,
        max_restarts=parameters.max_restarts, sort_function=WHICH_TO_SORT[parameters.which],
        p=parameters.p, stats=stats, **solver_kw)
    elapsed = time.perf_counter() - t0
    vals, S = np.linalg.eig(T)
    vecs = Q @ S
    order = WHICH_TO_SORT[parameters.which](vals)
    return vals[order], vecs[:, order], Statistics(
        elapsed, np.dtype(np.float64 if stats.get("arithmetic") == "real" else np.complex128),
        int(stats["matvecs"]), int(np.max(history.restarts)),
        int(np.max(history.matvecs)))


def arpack_eig(A, parameters: EigensolverParameters):
    op = MatvecCounter(A)
    t0 = time.perf_counter()
    vals, vecs = eigs(op, k=parameters.nev, which=parameters.which, ncv=parameters.ncv, tol=parameters.tol,
                      maxiter=parameters.max_restarts)
    elapsed = time.perf_counter() - t0
    order = WHICH_TO_SORT[parameters.which](vals)
    n_iters = (op.matvecs - parameters.ncv) // max(parameters.ncv - parameters.nev, 1)
    return vals[order], vecs[:, order], Statistics(elapsed, op.dtype, op.matvecs, n_iters, op.matvecs)


def compare(A, parameters: EigensolverParameters, verbose=False, **solver_kw):
    """One comparison: ARPACK vs the device Krylov-Schur solver.  Returns a list of CSV rows
    (one per method) with a ``match`` flag (eigenvalues equal to ``rtol = tol`` after matching)
    and the worst normalised residual of each method."""
    a_vals, a_vecs, a_stats = arpack_eig(A, parameters)
    k_vals, k_vecs, k_stats = krylov_schur_eig(A, parameters, **solver_kw)
    x, y = find_best_matching(a_vals, k_vals)
    match = bool(np.allclose(y, x, rtol=parameters.tol, atol=0.0))
    rows = []
    for method, vals, vecs, st in (("arpack", a_vals, a_vecs, a_stats), ("krylov-schur-mi355x", k_vals, k_vecs, k_stats)):
        _, rel = true_residuals(A, vals, vecs)
        rows.append({
            "method": method, "dtype": str(st.dtype), "nev": parameters.nev, "ncv": parameters.ncv,
            "tol": parameters.tol, "max_restarts": parameters.max_restarts, "p": parameters.p,
            "which": parameters.which, "elapsed": st.elapsed, "matvecs": st.matvecs, "restarts": st.restarts,
            "match": match, "max_rel_residual": float(rel.max()),
        })
        if verbose:
            print(residual_report(method, A, vals, vecs))
    if verbose:
        print(f"  ARPACK:        {a_stats.matvecs} matvecs in {a_stats.restarts} iterations ({a_stats.elapsed:.3f}s)")
        print(f"  partial_schur: {k_stats.matvecs} matvecs in {k_stats.restarts} restarts  ({k_stats.elapsed:.3f}s)"
              f"  match={match}")
    return rows


def sweep(A, out_csv, grid=None, whichs=("LM", "LR"), tol=1e-8, max_restarts=100_000, verbose=False, **solver_kw):
    """CSV over the (nev, ncv, p) x which grid (default: the reference's stress grid)."""
    grid = STRESS_GRID if grid is None else grid
    all_rows = []
    for which in whichs:
        for nev, ncv, p in grid:
            all_rows.extend(compare(A, EigensolverParameters(nev, ncv, tol, max_restarts, p, which), verbose,
                                    **solver_kw))
    with open(out_csv, "wt", newline="") as fp:
        writer = csv.DictWriter(fp, fieldnames=list(all_rows[0].keys()))
        writer.writeheader()
        writer.writerows(all_rows)
    return all_rows


def _is_matlab_v73(name):
    """MATLAB v7.3 files are HDF5 containers behind a 512-byte text header (``scipy.io.loadmat`` refuses
    them); the large SuiteSparse matrices -- af_shell10 among them -- are distributed in this format."""
    with open(name, "rb") as f:
        return f.read(19) == b"MATLAB 7.3 MAT-file"


def _load_mat_v73(name, h5py=None):
    """``Problem.A`` (or the first sparse variable) of a MATLAB v7.3 file: a sparse matrix is an HDF5 group with
    the CSC arrays ``data``, ``ir`` (row indices), ``jc`` (column pointers) and the attribute ``MATLAB_sparse``
    (= number of rows).  Needs h5py."""
    if h5py is None:
        try:
            import h5py
        except ImportError as e:
            raise ValueError(f"{name!r} is a MATLAB v7.3 (HDF5) file and h5py is not installed; "
                             "convert it to MatrixMarket (.mtx) or SciPy .npz") from e

    def sparse_groups(group, depth=0):
        if "MATLAB_sparse" in getattr(group, "attrs", {}):
            yield group
        elif hasattr(group, "keys") and depth < 3:
            keys = list(group.keys())
            for key in sorted(keys, key=lambda k: (k != "Problem", k != "A", k)):   # Problem.A first
                if not str(key).startswith("#"):
                    yield from sparse_groups(group[key], depth + 1)

    with h5py.File(name, "r") as f:
        for g in sparse_groups(f):
            n_rows = int(g.attrs["MATLAB_sparse"])
            jc = np.asarray(g["jc"], dtype=np.int64)
            ir = np.asarray(g["ir"], dtype=np.int64) if "ir" in g else np.zeros(0, np.int64)
            data = np.asarray(g["data"]) if "data" in g else np.zeros(0)
            if data.dtype.names:                                     # complex: compound (real, imag)
                data = data[data.dtype.names[0]] + 1j * data[data.dtype.names[1]]
            return sp.csc_matrix((data, ir, jc), shape=(n_rows, len(jc) - 1))
    raise ValueError(f"No sparse matrix found in {name!r}")


def load_matrix(path) -> sp.csr_matrix:
    """Square sparse matrix from a file, as canonical CSR.

    ``.mat``  SuiteSparse MATLAB layout: struct ``Problem`` with field ``A``
              (scripts/utils.py:102-116 of the reference); any top-level sparse variable otherwise;
              MATLAB v7.3 (HDF5) files through h5py when it is installed, a clear error otherwise;
    ``.mtx`` / ``.mtx.gz``  MatrixMarket;   ``.npz``  ``scipy.sparse.save_npz``.
    """
    name = os.fspath(path)
    if name.endswith(".mat") and _is_matlab_v73(name):
        A = _load_mat_v73(name)
    elif name.endswith(".mat"):
        data = scipy.io.loadmat(name, squeeze_me=False)
        A = None
        prob = data.get("Problem")
        if prob is not None:
            cand = prob["A"][0, 0]
            if sp.issparse(cand):
                A = cand
        if A is None:
            for key, val in data.items():
                if not key.startswith("__") and sp.issparse(val):
                    A = val
                    break
        if A is None:
            raise ValueError(f"No sparse matrix found in {name!r}")
    elif name.endswith((".mtx", ".mtx.gz")):
        A = scipy.io.mmread(name)
    elif name.endswith(".npz"):
        A = sp.load_npz(name)
    else:
        raise ValueError(f"unknown matrix file type: {name!r}")
    A = sp.csr_matrix(A)
    A.sum_duplicates()
    A.sort_indices()
    if A.shape[0] != A.shape[1]:
        raise ValueError(f"matrix in {name!r} is not square: {A.shape}")
    return A
