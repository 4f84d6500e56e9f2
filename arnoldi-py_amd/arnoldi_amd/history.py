"""Per-eigenvalue convergence bookkeeping returned by ``partial_schur``.

Same fields and helpers as the reference's dataclass
(src/arnoldi/explicit_restarts.py:13-28).
"""
import dataclasses

import numpy as np


@dataclasses.dataclass
class History:
    matvecs: np.ndarray   # int32[k], the reference's (over-)estimate, krylov_schur.py:63
    restarts: np.ndarray  # int32[k], restart index + 1 at which entry k was last under tol

    @classmethod
    def from_k(cls, k):
        z = np.zeros(k, dtype=np.int32)
        return cls(z, z.copy())

    @property
    def k(self):
        return self.matvecs.shape[0]

    @property
    def total_matvecs(self):
        return self.matvecs.sum()
