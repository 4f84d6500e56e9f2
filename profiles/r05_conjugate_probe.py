#!/usr/bin/env python3
"""Why the HIP path's early-stopped Markov solve came out as the complex CONJUGATE of the reference's (tests, round 5):
for a real matrix and the reference's real start vector the projected matrix H is exactly real, every complex Schur form of
it has a mirror image, and which one zgees returns hangs on the SIGNS OF THE ZERO imaginary parts of H.  Compare, on this
machine: the oracle (the reference's BLAS calls) and the HIP path -- diag(T), and the signed zeros of H after the first
expansion."""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "arnoldi-py_amd")
import oracle  # noqa: E402
from arnoldi_amd import matrices, partial_schur  # noqa: E402
from arnoldi_amd.engine import ArnoldiContext, as_operator  # noqa: E402
from arnoldi_amd.utils import arg_largest_real  # noqa: E402

A = matrices.mark(300)
n, m = A.shape[0], 20
for tol in (0.3, 0.2):
    np.random.seed(0)
    Qo, To, ho = oracle.krylov_schur(A, 5, max_dim=m, stopping_criterion=tol, sort_function=oracle.arg_largest_real)
    np.random.seed(0)
    Q, T, h = partial_schur(A, 5, max_dim=m, stopping_criterion=tol, sort_function=arg_largest_real)
    print(f"tol {tol}: restarts oracle {ho.restarts.max()} hip {h.restarts.max()}")
    print("  oracle diag(T)", np.round(np.diag(To), 6))
    print("  hip    diag(T)", np.round(np.diag(T), 6))
    print("  |T - To| max", float(np.abs(np.diag(T) - np.diag(To)).max()), " |T - conj(To)| max", float(np.abs(np.diag(T) - np.conj(np.diag(To))).max()))
np.random.seed(0)
v0 = oracle.random_unit_vector(n, np.complex128)
Vo = np.zeros((n, m + 1), np.complex128, order="F")
Ho = np.zeros((m + 1, m), np.complex128)
Vo[:, 0] = v0
oracle.arnoldi_expand(A, Vo, Ho, 1e-8)
ctx = ArnoldiContext(as_operator(A), m)
ctx.set_start_vector(v0)
H = np.zeros((m + 1, m), np.complex128)
ctx.expand(H, 0, m, 1e-8)
neg = lambda M: int(np.signbit(M.imag[np.triu_indices(m, -1, m)[0], np.triu_indices(m, -1, m)[1]] if False else M.imag).sum())   # noqa: E731
print("first expansion: max |H - Ho|", float(np.abs(H - Ho).max()), " imag parts all zero:", not H.imag.any(), not Ho.imag.any())
print("  entries with imag == -0.0:  hip", neg(H), " oracle", neg(Ho), " of", H.size)
import scipy.linalg  # noqa: E402

for name, M in (("hip H", H[:m, :m]), ("oracle H", Ho[:m, :m]), ("hip H with +0 imag", H[:m, :m].real + 0j), ("oracle H with +0 imag", Ho[:m, :m].real + 0j)):
    Ts, _ = scipy.linalg.schur(M, output="complex")
    d = np.diag(Ts)
    lead = d[np.argsort(-d.real)][:6]
    print(f"  zgees({name}): leading eigenvalues by real part {np.round(lead, 5)}")
