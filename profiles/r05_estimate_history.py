#!/usr/bin/env python3
"""Per-restart convergence estimates (krylov_schur.py:91-92) of configs 2 and 4 on the HIP path, to choose the loose
``stopping_criterion`` of tests/golden/make_golden_large.py c2 / c4: the iteration does not depend on the criterion until
it stops, so the criterion is read off this history.  Output: profiles/r05_estimate_history.txt."""
import sys
import time

import numpy as np

sys.path.insert(0, "arnoldi-py_amd")
from arnoldi_amd import matrices  # noqa: E402
from arnoldi_amd.krylov_schur import KrylovSchurSolver  # noqa: E402
from arnoldi_amd.utils import arg_largest_magnitude  # noqa: E402


def history(name, A, nev, m, p, restarts, sort=arg_largest_magnitude):
    np.random.seed(0)
    s = KrylovSchurSolver(A, nev, m, p, 1e-300, sort)
    t0 = time.time()
    s.start()
    for r in range(restarts):
        s.contract(r)
        print(f"{name} restart {r + 1:3d}  max estimate[:nev] {s.estimate.max():.6e}  min {s.estimate.min():.3e}", flush=True)
        s.expand()
    print(f"{name}: {restarts} restarts in {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "markov":          # the README's matrix at n = 10M, sorted LR (north_star: "synthetic Markov")
        from arnoldi_amd.utils import arg_largest_real

        history("markov10m", matrices.mark(4472), 5, 20, 10, 40, arg_largest_real)
        sys.exit(0)
    history("c2", matrices.laplace2d(1000, 1001), 10, 40, 15, 60)
    history("c4", matrices.laplace3d(251, 252, 253), 10, 40, 15, 12)
