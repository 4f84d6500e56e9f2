"""Host-side helpers of the Krylov-Schur driver (the O(m^3) part stays on the CPU by
design: north_star "the small Hessenberg Schur/restart step left on the host via LAPACK").

Interface mirror of src/arnoldi/utils.py: ``rand_normalized_vector`` (:7-13),
``arg_largest_magnitude`` (:16-17), ``arg_largest_real`` (:20-21),
``ordered_schur`` (:32-67).
"""
from __future__ import annotations

import contextlib
import os
import threading
import warnings

import numpy as np
import scipy.linalg
from scipy.linalg import lapack

__all__ = [
    "rand_normalized_vector",
    "StartVector",
    "arg_largest_magnitude",
    "arg_largest_real",
    "ordered_schur",
    "reorder_schur",
    "complex_schur",
]


_blas_lock = threading.Lock()
_blas_users = 0            # solves currently inside host_blas_threads(), over all host threads
_blas_limit = None         # the one threadpool_limits object they share
_blas_warned = False


def _blas_thread_setting():
    """AKS_HOST_BLAS_THREADS: "keep" -> None, a positive integer -> that limit, unset -> 1; anything else is
    rejected here, once, with a clear message (not as an int() traceback inside every solver)."""
    want = os.environ.get("AKS_HOST_BLAS_THREADS", "1").strip()
    if want == "keep":
        return None
    try:
        k = int(want)
    except ValueError:
        k = 0
    if k < 1:
        raise ValueError(f"AKS_HOST_BLAS_THREADS={want!r}: expected 'keep' or a positive integer")
    return k


@contextlib.contextmanager
def host_blas_threads():
    """The host side of a restart is LAPACK on an ``m x m`` matrix (m <= 128: Schur form, reordering, a few
    products) between two waits for the device.  A threaded BLAS makes that *slower*, and not by a little: its
    pool (one thread per visible CPU -- 256 on the 8-GPU hosts) is woken for a few microseconds of work and then
    spins, which on a box with a CPU quota stalls the thread that waits for the GPU.  Measured on MI355X hosts
    (profiles/r03_host_gap.txt, random CSR n = 1.25M, same build, process after process): 4.1 - 4.6 ms per restart
    with the default pools, 2.39 ms with one thread -- the whole difference sits in the wait for H or in the
    Schur call.  The solvers therefore run their loop inside this context: BLAS pools limited to ONE thread
    (threadpoolctl, restored on exit).  ``AKS_HOST_BLAS_THREADS=keep`` leaves the pools alone, an integer sets
    another limit.

    The BLAS pool is one per PROCESS while solves may run on several host threads at once (device.py keeps its
    stream per thread), so the limit is reference-counted: the first solve to enter sets it, the last one to leave
    restores the pools -- interleaved enters and exits can no longer leave the pool stuck at one thread
    (ADVICE r03).  threadpoolctl is an optional dependency: without it nothing is changed and a warning says so
    once, because small problems then run up to 2x slower on many-core hosts."""
    global _blas_users, _blas_limit, _blas_warned
    limit = _blas_thread_setting()
    if limit is None:
        yield
        return
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        if not _blas_warned:
            _blas_warned = True
            warnings.warn("threadpoolctl is not installed: the host BLAS pools keep their threads inside a solve "
                          "(up to 2x slower restarts on small problems, profiles/r03_host_gap.txt)", RuntimeWarning,
                          stacklevel=3)
        yield
        return
    with _blas_lock:
        if _blas_users == 0:
            _blas_limit = threadpool_limits(limits=limit, user_api="blas")
        _blas_users += 1
    try:
        yield
    finally:
        with _blas_lock:
            _blas_users -= 1
            if _blas_users == 0 and _blas_limit is not None:
                _blas_limit.restore_original_limits()
                _blas_limit = None


_NATIVE_RANDN_FROM = 1_000_000      # below this NumPy's own loop is a few milliseconds
_native_randn = None                # ctypes function, False once known to be unavailable
_randn_lock = threading.Lock()      # one native draw at a time: it reads, advances and writes back the GLOBAL generator


def _numpy_generator_lock():
    """The lock NumPy's own ``randn`` holds for its whole call (the global RandomState's bit-generator lock), or None."""
    try:
        return np.random.mtrand._rand._bit_generator.lock
    except AttributeError:
        return None


def legacy_randn(n):
    """``np.random.randn(n)`` -- the same bits, the same generator state afterwards -- computed by
    ``aks_legacy_randn`` (csrc/aks_host_rng.cpp: the Mersenne Twister sequentially, the polar-method arithmetic on host
    threads) when n is large: 3-4 x faster than NumPy's single-threaded loop at n = 10M, where that draw is the longest
    single piece of a whole ``partial_schur`` call.  Falls back to NumPy itself -- the reference
    implementation of this stream, not a stand-in for device work -- when the library (or a build of it without the helper:
    tests/mock_rccl) does not export the symbol, when the global generator is not the legacy MT19937, or on any error."""
    global _native_randn
    n = int(n)
    if n < _NATIVE_RANDN_FROM or _native_randn is False or os.environ.get("AKS_NATIVE_RANDN", "1") == "0":
        return np.random.randn(n)
    if _native_randn is None:
        try:
            import ctypes as C

            from . import _hip

            fn = C.CDLL(_hip.LIB_PATH).aks_legacy_randn
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p, C.c_int64]
            _native_randn = fn
        except (OSError, AttributeError):
            _native_randn = False
            return np.random.randn(n)
    import ctypes as C

    # get_state .. native draw (GIL released) .. set_state is one critical section: two solver constructors
    # starting at once would otherwise read the same MT state and draw the same vector, and a ``np.random`` draw by any
    # other thread during the native call would be overwritten by set_state.  NumPy's randn holds the generator's lock
    # for its whole call; so does this (get_state / set_state themselves do not take it).
    glock = _numpy_generator_lock()
    if glock is None:
        return np.random.randn(n)
    with _randn_lock, glock:
        state = np.random.get_state()
        failed = state[0] != "MT19937"
        if not failed:
            key = np.array(state[1], dtype=np.uint32)                 # (a copy: the saved state stays intact)
            pos, has_gauss, gauss = C.c_int32(int(state[2])), C.c_int32(int(state[3])), C.c_double(float(state[4]))
            out = np.empty(n, np.float64)
            rc = _native_randn(key.ctypes.data, C.byref(pos), C.byref(has_gauss), C.byref(gauss), out.ctypes.data, n)
            failed = rc != 0
            if failed:
                np.random.set_state(state)
            else:
                np.random.set_state(("MT19937", key, pos.value, has_gauss.value, gauss.value))
    return np.random.randn(n) if failed else out


def rand_normalized_vector(n, dtype=np.float64):
    """Unit 2-norm vector of ``n`` standard normals drawn from NumPy's *global* legacy
    generator, so ``np.random.seed(s)`` gives the reference's start vector bit for bit."""
    draws = legacy_randn(n)
    if n >= _NATIVE_RANDN_FROM and np.dtype(dtype) == np.complex128:
        # The reference's two statements (utils.py:10-11: astype, then ``v /= norm``) with the complex division spelled
        # out: NumPy divides a complex array by a real scalar with Smith's formula, which for a divisor c + 0i is
        # (a + b*0) * (1/c) and (b - a*0) * (1/c) -- for b = 0 the real part times the reciprocal and +0.  Same bits
        # (tests/test_host_logic.py), a third of the time of the generic complex division at n = 10M.
        vec = np.empty(n, np.complex128)
        vec.real = draws
        vec.imag = 0.0
        norm = np.linalg.norm(vec)
        re = vec.real
        np.multiply(re, 1.0 / norm, out=re)
        return vec
    vec = draws.astype(dtype)
    vec /= np.linalg.norm(vec)
    return vec


class StartVector:
    """``rand_normalized_vector(n, dtype)`` drawn on a helper thread while the caller sets the operator up.

    At n = 10M the reference's draw (legacy ``randn``: one generator, one thread, 0.17-0.28 s) is as long as the whole
    device-side set-up and three times the solve; NumPy's legacy generator fills its array with the GIL released, and the
    operator set-up is C code behind ctypes plus device copies, so the two overlap.  The draw is the same single
    ``np.random.randn(n)`` call on the global generator -- the reference's start vector bit for bit -- and nothing else
    in the set-up touches that generator.  Below ``ASYNC_FROM`` entries (and for a given ``v0``) no thread is started."""

    ASYNC_FROM = 1_000_000

    def __init__(self, n, dtype, v0=None):
        self._value, self._error, self._thread = None, None, None
        if v0 is not None:
            self._value = v0
        elif n < self.ASYNC_FROM:
            self._value = rand_normalized_vector(n, dtype)
        else:
            def draw():
                try:
                    self._value = rand_normalized_vector(n, dtype)
                except BaseException as e:  # noqa: BLE001  (re-raised in get())
                    self._error = e

            self._thread = threading.Thread(target=draw, name="aks-start-vector", daemon=True)
            self._thread.start()

    def get(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._error is not None:
            raise self._error
        return self._value


def arg_largest_magnitude(x):
    """Permutation that lists ``x`` by decreasing modulus ("LM")."""
    return np.argsort(-np.abs(x))


def arg_largest_real(x):
    """Permutation that lists ``x`` by decreasing real part ("LR")."""
    return np.argsort(-np.real(x))


_REAL_SCHUR_RETRY = 8       # after a real Schur form with 2 x 2 blocks: try the real route again at the 8th call


def complex_schur(a, memo=None):
    """``(T, Z)`` with ``a = Z T Z^H``, T upper triangular, both complex128 -- what the reference gets from
    ``scipy.linalg.schur(a, output="complex")`` (``zgees``; krylov_schur.py:69, utils.py:45).

    Shortcut (round 4): the projected matrix of a REAL operator started from a real vector is exactly real -- and stays
    so from restart to restart as long as the Schur vectors are (real SpMV, real projections, a real ``Qp`` in the
    compression keep every imaginary part of V and H at +-0).  Its real Schur form (``dgees``) costs a third of the
    complex one -- 0.07 vs 0.17 ms at m = 20, 0.32 vs 0.89 ms at m = 40 -- and that call is what the device waits for
    between two restarts once the shards are small (at m = 40 the whole idle gap of a restart).  Whenever that
    real Schur form is TRIANGULAR (all eigenvalues real: Laplacians, the Markov chain) it IS a complex Schur form of
    ``a``, and it is returned as such.  A real matrix with complex pairs (2x2 blocks) and every complex matrix go
    through ``zgees`` as before.  The Schur form is unique up to the phases of the Schur vectors only, so ``(T, Z)`` may
    differ from ``zgees``' output by such phases; the invariant subspaces -- all the Krylov-Schur iteration uses -- and
    hence History, eigenvalues and residuals do not.  ``AKS_REAL_SCHUR=0`` switches the shortcut off.

    ``memo`` (a dict the caller keeps per solve): once the real attempt has produced 2 x 2 blocks it is skipped for the
    next ``_REAL_SCHUR_RETRY - 1`` calls -- a real operator with complex Ritz pairs would otherwise pay dgees AND zgees
    while H stays exactly real."""
    a = np.asarray(a)
    if np.iscomplexobj(a) and os.environ.get("AKS_REAL_SCHUR", "1") != "0" and not a.imag.any():
        if memo is not None and memo.get("skip_real", 0) > 0:
            memo["skip_real"] -= 1
        else:
            Tr, Zr = scipy.linalg.schur(np.ascontiguousarray(a.real), output="real")
            if not np.diagonal(Tr, -1).any():              # no 2 x 2 block: a triangular, i.e. complex, Schur form
                return Tr.astype(np.complex128), Zr.astype(np.complex128)
            if memo is not None:
                memo["skip_real"] = _REAL_SCHUR_RETRY - 1
                memo["real_attempts_wasted"] = memo.get("real_attempts_wasted", 0) + 1
    return scipy.linalg.schur(a, output="complex")


_SWAPPERS = {"F": lapack.ctrexc, "D": lapack.ztrexc, "f": lapack.strexc, "d": lapack.dtrexc}


def reorder_schur(T, Z, order):
    """Permute the diagonal of an upper-triangular (complex) Schur factor.

    ``order[i]`` is the current diagonal position of the eigenvalue that must end up
    at position ``i``.  Each misplaced eigenvalue is moved by one LAPACK ``?trexc``
    call (1-based indices), front to back, exactly the sequence of unitary swaps the
    reference performs (src/arnoldi/utils.py:49-63), so the resulting (T, Z) agree
    with it to rounding and not merely up to a unitary change of basis.
    """
    swap = _SWAPPERS[np.asarray(T).dtype.char]
    position = list(range(T.shape[0]))  # position[k] = original label sitting at slot k
    for slot, label in enumerate(order):
        at = position.index(label)
        if at == slot:
            continue
        T, Z, info = swap(T, Z, at + 1, slot + 1)
        if info != 0:
            raise np.linalg.LinAlgError(f"?trexc failed with info={info}")
        position.insert(slot, position.pop(at))
    return T, Z


def ordered_schur(a, output="real", *, sort_function=None):
    """Schur decomposition ``a = Z T Z^H`` whose eigenvalues appear on ``diag(T)`` in the
    order given by ``sort_function`` (default: largest magnitude first).

    ``output="complex"`` is the reference's (utils.py:32-67).  ``output="real"`` -- "not implemented yet" there
    (utils.py:64-65; its test is an xfail, tests/test_utils.py:51-87) -- gives the REAL Schur form of a real matrix:
    orthogonal ``Z``, quasi-triangular ``T`` with 1x1 and 2x2 diagonal blocks in the order of ``sort_function``
    (a conjugate pair takes the better rank of its two members and moves as one block; ``?trexc``), the form the
    real-arithmetic solver works with (krylov_schur_real.py).
    """
    if sort_function is None:
        sort_function = arg_largest_magnitude
    if output == "complex":
        T, Z = scipy.linalg.schur(a, output=output)
        return reorder_schur(T, Z, sort_function(np.diag(T)))
    if output != "real":
        raise ValueError("output must be 'complex' or 'real'")
    if np.iscomplexobj(a):
        raise ValueError("output='real' needs a real matrix")
    from .krylov_schur_real import reorder_real_schur

    T, Z = scipy.linalg.schur(a, output="real")
    return reorder_real_schur(T, Z, sort_function)
