"""MI355X-native drop-in for ``arnoldi.partial_schur`` (cournape/arnoldi-py).

    import arnoldi_amd as arnoldi
    Q, T, history = arnoldi.partial_schur(A, nev, max_dim=..., sort_function=...)

Same signature, defaults, return types and exceptions as the reference
(src/arnoldi/__init__.py:1-3, src/arnoldi/krylov_schur.py:10-114); the O(n) work runs
in hand-written HIP kernels for gfx950 behind ``libarnoldi_hip.so``.
"""
from ._version import __version__  # noqa: F401
from .krylov_schur import partial_schur  # noqa: F401
