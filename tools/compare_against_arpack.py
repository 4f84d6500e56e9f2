#!/usr/bin/env python3
"""Compare the MI355X Krylov-Schur solver with SciPy's ARPACK on one matrix, or sweep the
reference's (nev, ncv, p) stress grid into a CSV -- counterpart of the reference's
scripts/compare-against-arpack.py and scripts/stress-test.py (SLEPc leg omitted).

    python tools/compare_against_arpack.py mark:50 --nev 5 --ncv 20 --which LR
    python tools/compare_against_arpack.py path/to/af_shell10.mat --sweep out.csv
    python tools/compare_against_arpack.py laplace2d:300x301 --nev 10 --ncv 40

Matrix argument: a .mat / .mtx / .npz file, or  mark:M | laplace2d:NXxNY | laplace3d:NXxNYxNZ |
random:N[:per_row] | banded:N[:per_row] | shell:NXxNY (5 unknowns per node).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))

import numpy as np  # noqa: E402

from arnoldi_amd import harness, matrices  # noqa: E402


def build(spec):
    if os.path.exists(spec):
        return harness.load_matrix(spec)
    kind, _, arg = spec.partition(":")
    if kind == "mark":
        return matrices.mark(int(arg))
    if kind == "laplace2d":
        return matrices.laplace2d(*(int(v) for v in arg.split("x")))
    if kind == "laplace3d":
        return matrices.laplace3d(*(int(v) for v in arg.split("x")))
    if kind == "shell":
        return matrices.shell_csr(*(int(v) for v in arg.split("x")))
    if kind in ("random", "banded"):
        parts = [int(v) for v in arg.split(":")]
        gen = matrices.random_csr if kind == "random" else matrices.banded_csr
        return gen(parts[0], *(parts[1:2] or ([5] if kind == "random" else [35])))
    raise SystemExit(f"cannot interpret matrix argument {spec!r}")


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("matrix")
    ap.add_argument("--nev", type=int, default=6)
    ap.add_argument("--ncv", type=int, default=20)
    ap.add_argument("--p", type=int, default=None)
    ap.add_argument("--which", choices=["LM", "LR"], default="LR")
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--max-it", type=int, default=100_000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--sweep", metavar="CSV", help="run the whole stress grid and write this CSV")
    ap.add_argument("--arithmetic", choices=["complex", "real", "auto"], default="complex",
                    help="complex = the reference's iteration (default); real = real Schur form on a real-packed "
                         "basis (real matrices only)")
    args = ap.parse_args()

    A = build(args.matrix)
    print(f"Matrix: {args.matrix}  shape={A.shape[0]}x{A.shape[1]}, nnz={A.nnz}, dtype={A.dtype}")
    np.random.seed(args.seed)
    if args.sweep:
        rows = harness.sweep(A, args.sweep, tol=args.tol, max_restarts=args.max_it, verbose=True,
                             arithmetic=args.arithmetic)
        bad = [r for r in rows if not r["match"]]
        print(f"wrote {args.sweep}: {len(rows)} rows, {len(bad) // 2} parameter sets without eigenvalue match")
        return 1 if bad else 0
    params = harness.EigensolverParameters(args.nev, args.ncv, args.tol, args.max_it, args.p, args.which)
    rows = harness.compare(A, params, verbose=True, arithmetic=args.arithmetic)
    return 0 if rows[0]["match"] else 1


if __name__ == "__main__":
    sys.exit(main())
