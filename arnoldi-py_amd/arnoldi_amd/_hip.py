"""ctypes binding of libarnoldi_hip.so (the C ABI in include/arnoldi_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this
module raises.  Build with ``make -C arnoldi-py_amd`` (or
``python -c "import __graft_entry__ as g; g.build()"``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

ABI_VERSION = 6
MAX_DIM = 128          # AKS_MAX_DIM
MAX_TRUNC = 96         # AKS_MAX_TRUNC
SPMV_TILE_NNZ = 256    # AKS_SPMV_TILE_NNZ

LIB_PATH = os.environ.get(  # AKS_LIB_PATH: A/B a differently built library (profiles/ab_kernels.py)
    "AKS_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libarnoldi_hip.so"))


class HipLibraryError(RuntimeError):
    """libarnoldi_hip.so is missing, stale, or returned an error status."""


class WsLayout(C.Structure):
    _fields_ = [
        ("total_bytes", C.c_int64),
        ("ctrl_off", C.c_int64),
        ("red1_off", C.c_int64),
        ("red2_off", C.c_int64),
        ("red3_off", C.c_int64),
        ("partial_off", C.c_int64),
        ("n_blocks", C.c_int32),
        ("ld_partial", C.c_int32),
        ("red_len", C.c_int32),
        ("pad_", C.c_int32),
        ("colscale_off", C.c_int64),
    ]


class Ctrl(C.Structure):
    """Mirror of ``aks_ctrl`` (64 bytes)."""

    _fields_ = [
        ("broken", C.c_int32),
        ("n_iter", C.c_int32),
        ("steps_done", C.c_int32),
        ("second_passes", C.c_int32),
        ("beta_in", C.c_double),
        ("beta", C.c_double),
        ("real_mode", C.c_int32),
        ("deferred", C.c_int32),
        ("ticket", C.c_uint32 * 4),
        ("reserved", C.c_double),
    ]


_P = C.c_void_p
_I32, _I64, _F64 = C.c_int32, C.c_int64, C.c_double

PB_SLAB_BITS = 13       # AKS_PB_SLAB_BITS
PB_ROWBLOCK_BITS = 13   # AKS_PB_ROWBLOCK_BITS
PB_RUNS_PER_ROUND = 32  # AKS_PB_WAVES * AKS_PB_RUNS_PER_WAVE
EXPAND_FROM_W, EXPAND_REAL_PACKED, EXPAND_LAZY_THIRD, EXPAND_DEFER_SCALE = 1, 2, 4, 8   # AKS_EXPAND_* flags
COMM_ID_BYTES = 128     # AKS_COMM_ID_BYTES


class PbRun(C.Structure):
    """Mirror of ``aks_pb_run`` (16 bytes)."""

    _fields_ = [("start0", C.c_uint32), ("start1", C.c_uint32), ("start2", C.c_uint32), ("info", C.c_uint32)]


class PbSizes(C.Structure):
    """Mirror of ``aks_pb_sizes``."""

    _fields_ = [("nnz_pad", _I64), ("n_runs", _I64), ("n_lrow", _I64), ("n_slabs", _I32), ("n_rowblocks", _I32)]


class PbPlanArrays(C.Structure):
    """Mirror of ``aks_pb_plan_arrays`` (host pointers into a plan, valid until it is destroyed)."""

    _fields_ = [(name, _P) for name in ("val", "lcol", "slab_begin", "slab_end", "runs", "rb_run_ptr", "lrow")]


class PbMatrix(C.Structure):
    """Mirror of ``aks_pb_matrix`` (device pointers of the tile-binned SpMV form)."""

    _fields_ = [
        ("n_rows", _I64), ("n_cols", _I64), ("nnz", _I64), ("nnz_pad", _I64), ("n_runs", _I64), ("n_lrow", _I64),
        ("n_slabs", _I32), ("n_rowblocks", _I32), ("values_complex", _I32), ("pad_", _I32),
        ("d_val", _P), ("d_lcol", _P), ("d_slab_begin", _P), ("d_slab_end", _P), ("d_runs", _P),
        ("d_rb_run_ptr", _P), ("d_lrow", _P), ("d_prod", _P),
    ]


class SellMatrix(C.Structure):
    """Mirror of ``aks_sell_matrix`` (device pointers of the sliced SpMV form)."""

    _fields_ = [
        ("n_rows", _I64), ("n_cols", _I64), ("nnz", _I64), ("nnz_pad", _I64), ("n_slices", _I64),
        ("values_complex", _I32), ("pad_", _I32), ("d_slice_ptr", _P), ("d_col", _P), ("d_val", _P),
    ]


class CsrBlock(C.Structure):
    """Mirror of ``aks_csr_block``: one CSR block with its SpMV plan (any of the three forms)."""

    _fields_ = [
        ("n_rows", _I64), ("n_cols", _I64), ("d_indptr", _P), ("d_indices", _P), ("d_values", _P),
        ("d_tiles", _P), ("n_tiles", _I64), ("values_complex", _I32), ("lanes_per_row", _I32),
        ("pb", C.POINTER(PbMatrix)), ("sell", C.POINTER(SellMatrix)),
    ]


class Shard(C.Structure):
    """Mirror of ``aks_shard``: this rank's rows of the operator and its ghost exchange."""

    _fields_ = [
        ("diag", CsrBlock), ("off", CsrBlock), ("comm", _P), ("d_send_idx", _P), ("d_sendbuf", _P),
        ("d_ghostbuf", _P), ("n_send", _I64), ("n_ghost", _I64), ("send_counts", C.POINTER(_I64)),
        ("recv_counts", C.POINTER(_I64)), ("any_exchange", _I32), ("pad_", _I32),
    ]


# name -> (restype, argtypes); one entry per function declared in include/arnoldi_hip.h
SIGNATURES = {
    "aks_last_error": (C.c_char_p, []),
    "aks_abi_version": (_I32, []),
    "aks_workspace_layout": (C.c_int, [_I64, _I32, C.POINTER(WsLayout)]),
    "aks_workspace_init": (C.c_int, [_P, _I64, _I64, _I32, _P]),
    "aks_csr_plan_tiles": (_I64, [_P, _I64, _I32, _P, _I64]),
    "aks_csr_spmv": (C.c_int, [_I64, _P, _P, _P, _I32, _P, _I64, _I32, _P, _P, _I32, _P, _P]),
    "aks_gs_project": (C.c_int, [_I64, _I32, _P, _I64, _P, _P, _I64, _I32, _P]),
    "aks_gs_update_project": (C.c_int, [_I64, _I32, _P, _I64, _P, _P, _I64, _I32, _P]),
    "aks_gs_update_norm": (C.c_int, [_I64, _I32, _P, _I64, _P, _F64, _P, _I64, _I32, _P]),
    "aks_gs_finish": (C.c_int, [_I64, _I32, _P, _P, _I64, _F64, _F64, _I32, _P, _I64, _I32, _P]),
    "aks_dgks_gs": (C.c_int, [_I64, _I32, _P, _I64, _P, _P, _I64, _F64, _F64, _I32, _P, _I64, _I32, _P]),
    "aks_device_init": (C.c_int, []),
    "aks_pb_params": (C.c_int, [C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "aks_pb_plan_create": (_P, [_P, _P, _P, _I32, _I64, _I64, C.POINTER(PbSizes)]),
    "aks_pb_plan_export": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "aks_pb_plan_view": (C.c_int, [_P, C.POINTER(PbPlanArrays)]),
    "aks_pb_plan_destroy": (None, [_P]),
    "aks_pb_spmv": (C.c_int, [C.POINTER(PbMatrix), _P, _P, _I32, _P, _P]),
    "aks_sell_plan_size": (_I64, [_P, _I64]),
    "aks_sell_plan_fill": (C.c_int, [_P, _P, _P, _I32, _I64, _P, _P, _P]),
    "aks_sell_spmv": (C.c_int, [C.POINTER(SellMatrix), _P, _P, _I32, _P, _P]),
    "aks_arnoldi_expand": (C.c_int, [C.POINTER(Shard), _P, _I64, _P, _I64, _I32, _I32, _F64, _F64, _P, _I64, _I32, _P, _P,
                                     _I32]),
    "aks_shard_apply": (C.c_int, [C.POINTER(Shard), _P, _P, _P, _P, _I32]),
    "aks_comm_unique_id": (C.c_int, [_P]),
    "aks_comm_create": (C.c_int, [_P, _I32, _I32, C.POINTER(_P)]),
    "aks_comm_destroy": (C.c_int, [_P]),
    "aks_comm_allreduce_sum": (C.c_int, [_P, _P, _I64, _P]),
    "aks_comm_allreduce_path": (C.c_int, [_P, C.c_char_p, _I64]),
    "aks_comm_alltoallv": (C.c_int, [_P, _P, C.POINTER(_I64), C.POINTER(_I64), _P, C.POINTER(_I64), C.POINTER(_I64), _P]),
    "aks_comm_status": (C.c_int, [_P, C.c_char_p, _I64]),
    "aks_comm_graph_retain": (C.c_int, [_P]),
    "aks_comm_graph_release": (C.c_int, [_P]),
    "aks_stream_copy": (C.c_int, [_P, _P, _I64, _P]),
    "aks_runtime_versions": (C.c_int, [C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "aks_workspace_set_real": (C.c_int, [_P, _I32, _P]),
    "aks_csr_spmv_real": (C.c_int, [_I64, _P, _P, _P, _P, _I64, _I32, _P, _P, _I32, _P, _P]),
    "aks_pb_spmv_real": (C.c_int, [C.POINTER(PbMatrix), _P, _P, _I32, _P, _P]),
    "aks_sell_spmv_real": (C.c_int, [C.POINTER(SellMatrix), _P, _P, _I32, _P, _P]),
    "aks_gather_f64": (C.c_int, [_I64, _P, _P, _P, _P]),
    "aks_truncate": (C.c_int, [_I64, _I32, _I32, _P, _I64, _P, _P]),
    "aks_truncate_ws": (C.c_int, [_I64, _I32, _I32, _P, _I64, _P, _I32, _P, _I64, _I32, _P]),
    "aks_shard_apply_col": (C.c_int, [C.POINTER(Shard), _P, _I64, _I32, _P, _P, _I64, _I32, _P, _I32]),
    "aks_combine": (C.c_int, [_I64, _I32, _I32, _P, _I64, _P, _P, _I64, _P]),
    "aks_scale": (C.c_int, [_I64, _P, _F64, _F64, _P]),
    "aks_gather_c128": (C.c_int, [_I64, _P, _P, _P, _P]),
    "aks_probe_create": (C.c_int, [_I32, C.POINTER(_P)]),
    "aks_probe_destroy": (C.c_int, [_P]),
    "aks_probe_reset": (C.c_int, [_P]),
    "aks_probe_read": (C.c_int, [_P, _I32, C.POINTER(_I32), C.POINTER(_F64)]),
}

_lock = threading.Lock()
_lib = None


def load():
    """Load (once) and return the ctypes library with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `make -C arnoldi-py_amd`). There is no CPU fallback."
            )
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the machine
            raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise HipLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        if lib.aks_abi_version() != ABI_VERSION:
            raise HipLibraryError(
                f"ABI mismatch: library {lib.aks_abi_version()} vs binding {ABI_VERSION}; rebuild"
            )
        _lib = lib
    return _lib


def check(status, what):
    """Raise HipLibraryError for a negative status code."""
    if status < 0:
        msg = load().aks_last_error()
        raise HipLibraryError(f"{what} failed ({status}): {msg.decode() if msg else '?'}")
    return status


_devices_ready = set()
_device_lock = threading.Lock()


def device_init(index):
    """``aks_device_init`` once per device of this process: the library keeps no state of its own, so the host
    layer remembers which devices have had the dynamic-LDS limit of the large-LDS kernels raised."""
    if index in _devices_ready:
        return
    from . import mem

    lib = load()                     # (takes _lock itself: not inside the device lock below)
    with _device_lock:
        if index in _devices_ready:
            return
        with mem.device_ctx(index):
            check(lib.aks_device_init(), "aks_device_init")
        _devices_ready.add(index)


PROBE_SPMV, PROBE_ORTHO, PROBE_PACK, PROBE_EXCHANGE, PROBE_DIAG, PROBE_OFFDIAG, PROBE_ALLREDUCE = range(7)   # AKS_PROBE_*


class Probe:
    """hipEvent pairs recorded by ``aks_arnoldi_expand`` around every SpMV / orthogonalisation."""

    def __init__(self, capacity=4096):
        self.handle = _P()
        check(load().aks_probe_create(capacity, C.byref(self.handle)), "aks_probe_create")

    def reset(self):
        check(load().aks_probe_reset(self.handle), "aks_probe_reset")

    def read(self, tag):
        """(count, total_ms) of the pairs recorded with ``tag``; waits for the last event."""
        n, ms = _I32(0), _F64(0.0)
        check(load().aks_probe_read(self.handle, tag, C.byref(n), C.byref(ms)), "aks_probe_read")
        return n.value, ms.value

    def __del__(self):
        try:
            if self.handle:
                load().aks_probe_destroy(self.handle)
                self.handle = _P()
        except Exception:
            pass


def pb_params():
    """(sub-slab bits, row-block bits, runs per round) of the loaded library's tile-binned SpMV form."""
    a, b, c = _I32(0), _I32(0), _I32(0)
    check(load().aks_pb_params(C.byref(a), C.byref(b), C.byref(c)), "aks_pb_params")
    return a.value, b.value, c.value


def workspace_layout(n_rows, max_dim):
    lay = WsLayout()
    check(load().aks_workspace_layout(n_rows, max_dim, C.byref(lay)), "aks_workspace_layout")
    return lay


def runtime_versions():
    """``{"hip_runtime": .., "hip_driver": .., "rccl": ..}`` of what this process actually runs on (``hipRuntimeGetVersion``,
    ``hipDriverGetVersion``, ``ncclGetVersion`` of the librccl the library loaded; ``rccl`` is None while none is loaded).
    A torch process runs on torch's bundled HIP / RCCL, a torch-free one on the system's ROCm."""
    a, b, c = _I32(0), _I32(0), _I32(0)
    check(load().aks_runtime_versions(C.byref(a), C.byref(b), C.byref(c)), "aks_runtime_versions")
    return {"hip_runtime": a.value, "hip_driver": b.value, "rccl": None if c.value < 0 else c.value}


def comm_status(handle):
    """Raise if a one-shot reduction of this rank's communicator timed out (``aks_comm_status``: one host-memory read)."""
    why = C.create_string_buffer(256)
    if load().aks_comm_status(handle, why, 256) > 0:
        raise HipLibraryError(why.value.decode())
