"""CPU oracle for the Krylov-Schur hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``arnoldi-py_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and there only as the checker / the reported CPU baseline.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
reference (``/root/reference/src/arnoldi``) in the build container with
``tests/golden/make_golden.py``; ``tests/test_oracle_golden.py`` checks every
oracle function against them.
"""
from .ks_oracle import (  # noqa: F401
    History,
    arg_largest_magnitude,
    arg_largest_real,
    arnoldi_expand,
    csr_matvec,
    dgks_gs,
    eig_residuals,
    explicit_restarts_with_deflation,
    krylov_schur,
    laplace_1d,
    laplace_1d_eigen,
    mark_matrix,
    mgs,
    naive_explicit_restarts,
    ordered_schur,
    random_unit_vector,
    ritz_from_v_and_h,
)
