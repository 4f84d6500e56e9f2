"""Only ``History`` lives here, for import compatibility with the reference
(``from arnoldi.explicit_restarts import History``, src/arnoldi/explicit_restarts.py:13).
The explicit-restart solvers themselves are outside the hot path."""
from .history import History  # noqa: F401
