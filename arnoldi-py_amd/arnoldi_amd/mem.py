"""Device memory, streams and events of the host layer: through the HIP runtime alone (default) or through torch.

The C ABI of libarnoldi_hip.so takes raw device pointers and a ``hipStream_t``; what the Python layer needs around it is
small: allocate / zero / copy buffers, views of rows and sub-blocks, one stream, a few events, pinned staging memory.

``AKS_HOST_ALLOC=hip`` (the default)  ``HipArray`` below: hipMalloc / hipMemcpyAsync / hipMemsetAsync / hipHostMalloc /
                                    events through ctypes on libamdhip64.so.  torch is never imported: the drop-in
                                    needs what the reference needs -- numpy and scipy (the reference's
                                    dependencies, pyproject.toml:9-13) -- plus the HIP runtime, and it runs on the
                                    SYSTEM's ROCm (HIP 7.2 / RCCL 2.27 on this image), not on the older runtime a torch
                                    wheel bundles.  hipGraph replay through the runtime's capture API; row-sharded solves
                                    through ``dist.HostComm`` (TCP rendezvous + the library's communicator;
                                    ``AKS_COMM=host``).
``AKS_HOST_ALLOC=torch``            interop: torch tensors, torch's current stream -- the host layer then composes with
                                    whatever else the caller does in torch (a torch.distributed process group carries
                                    the multi-rank set-up, hipGraph replay uses torch's capture API, tests use torch to
                                    inspect device buffers; the CPU tests drive the host logic on CPU tensors).  The
                                    backend is chosen once per process, when this module is imported.

Both backends expose the same handful of functions, and their arrays the same handful of methods (``data_ptr``, basic
slicing, ``view``, ``copy_``, ``zero_``, ``cpu().numpy()``, ``item``), which is all device.py / engine.py use.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

def _default_backend():
    """The HIP runtime alone -- the reference's own dependency set plus ROCm -- whether or not torch is installed (round 6:
    VERDICT r05 item 4; until then torch was preferred whenever it could be imported)."""
    return "hip"


# hipGraph captures are serialised over the host threads of a process: entering a capture synchronises the device and
# (torch) empties the allocator's cache -- hipFree -- which invalidates a capture another thread has in flight
# ("operation failed due to a previous error during capture": seen once in ~6 runs of two concurrently solving threads,
# round 5).  A capture happens once per launch sequence; replays take no lock.
_capture_lock = threading.Lock()

BACKEND = os.environ.get("AKS_HOST_ALLOC") or _default_backend()
if BACKEND not in ("torch", "hip"):
    raise ValueError(f"AKS_HOST_ALLOC={BACKEND!r}: expected 'torch' or 'hip'")

# =============================================================================================== torch backend
if BACKEND == "torch":
    import torch

    c128, f64, u8, i32, i64 = torch.complex128, torch.float64, torch.uint8, torch.int32, torch.int64
    Event = torch.cuda.Event

    def gpu_available():
        return torch.cuda.is_available()

    def current_device():
        return torch.cuda.current_device()

    def device_count():
        return torch.cuda.device_count()

    def set_device(index):
        torch.cuda.set_device(int(index))

    def as_device(device):
        return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    def device_ctx(index):
        return torch.cuda.device(index)

    def stream_ptr():
        return torch.cuda.current_stream().cuda_stream

    def synchronize():
        torch.cuda.synchronize()

    def zeros(shape, dtype, device):
        return torch.zeros(shape, dtype=dtype, device=device)

    def empty(shape, dtype, device):
        return torch.empty(shape, dtype=dtype, device=device)

    def upload(a, device):
        """Host ndarray -> device array (synchronous for pageable memory)."""
        return torch.from_numpy(a).to(device)

    def host(a):
        """Host ndarray as the source of a ``copy_`` into a device array."""
        return torch.from_numpy(a)

    def pinned_empty(shape, dtype):
        return torch.empty(shape, dtype=dtype, pin_memory=True)

    class Graph:
        """A launch sequence captured once into a hipGraph and replayed on the current stream."""

        def __init__(self, enqueue):
            # torch.cuda.graph() spelled out, without its device-wide synchronize / gc / empty_cache on entry (the hipFree of
            # an emptied cache is what invalidates another thread's capture) and with the thread's current stream restored
            # whatever capture_end does: a failed capture must leave the caller able to launch eagerly on ITS stream
            self.g = torch.cuda.CUDAGraph()
            origin, side = torch.cuda.current_stream(), torch.cuda.Stream()
            with _capture_lock:
                side.wait_stream(origin)
                with torch.cuda.stream(side):
                    self.g.capture_begin(capture_error_mode="relaxed")
                    try:
                        enqueue()
                    finally:
                        self.g.capture_end()
                origin.wait_stream(side)

        def replay(self):
            self.g.replay()

        def destroy(self):
            """Destroy the executable graph NOW (not when the garbage collector gets to it): graphs that captured a
            communicator's operations must be gone before the communicator is (dist.Comm.close)."""
            if self.g is not None:
                self.g.reset()
                self.g = None

# =============================================================================================== HIP backend
else:
    c128, f64, u8, i32, i64 = (np.dtype(t) for t in (np.complex128, np.float64, np.uint8, np.int32, np.int64))
    _D2H, _H2D, _D2D = 2, 1, 3                       # hipMemcpyKind
    _tls = threading.local()

    def _rt():
        lib = getattr(_rt, "lib", None)
        if lib is None:
            for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
                try:
                    lib = C.CDLL(name, mode=C.RTLD_GLOBAL)
                    break
                except OSError:
                    lib = None
            if lib is None:
                raise RuntimeError("AKS_HOST_ALLOC=hip: libamdhip64.so not found")
            for fn in ("hipMalloc", "hipFree", "hipHostMalloc", "hipHostFree", "hipMemsetAsync", "hipMemset2DAsync",
                       "hipMemcpyAsync", "hipMemcpy2DAsync", "hipStreamCreateWithFlags", "hipStreamSynchronize",
                       "hipEventCreateWithFlags", "hipEventRecord", "hipEventSynchronize", "hipEventDestroy",
                       "hipEventElapsedTime", "hipGetDeviceCount", "hipGetDevice", "hipSetDevice", "hipDeviceSynchronize",
                       "hipStreamBeginCapture", "hipStreamEndCapture", "hipGraphInstantiate", "hipGraphLaunch",
                       "hipGraphDestroy", "hipGraphExecDestroy", "hipRuntimeGetVersion"):
                getattr(lib, fn).restype = C.c_int
            lib.hipGetErrorString.restype = C.c_char_p
            _rt.lib = lib
        return lib

    def _ck(status, what):
        if status != 0:
            raise RuntimeError(f"{what} failed: {_rt().hipGetErrorString(status).decode()}")

    class Device:
        type = "cuda"

        def __init__(self, index):
            self.index = int(index)

        def __repr__(self):
            return f"hip:{self.index}"

    def gpu_available():
        try:
            n = C.c_int(0)
            return _rt().hipGetDeviceCount(C.byref(n)) == 0 and n.value > 0
        except RuntimeError:
            return False

    def current_device():
        d = C.c_int(0)
        _ck(_rt().hipGetDevice(C.byref(d)), "hipGetDevice")
        return d.value

    def device_count():
        n = C.c_int(0)
        _ck(_rt().hipGetDeviceCount(C.byref(n)), "hipGetDeviceCount")
        return n.value

    def set_device(index):
        _ck(_rt().hipSetDevice(int(index)), "hipSetDevice")

    def as_device(device):
        if device is None:
            return Device(current_device())
        if isinstance(device, Device):
            return device
        if isinstance(device, int):
            return Device(device)
        text = str(device)
        return Device(int(text.split(":")[1]) if ":" in text else current_device())

    class device_ctx:
        def __init__(self, index):
            self.index = index

        def __enter__(self):
            self.prev = current_device()
            _ck(_rt().hipSetDevice(self.index), "hipSetDevice")

        def __exit__(self, *exc):
            _ck(_rt().hipSetDevice(self.prev), "hipSetDevice")
            return False

    def stream_ptr():
        """One non-blocking stream per host thread AND device (created on first use on the device that is current:
        a stream belongs to the device it was created on)."""
        streams = getattr(_tls, "streams", None)
        if streams is None:
            streams = _tls.streams = {}
        dev = current_device()
        s = streams.get(dev)
        if s is None:
            h = C.c_void_p()
            _ck(_rt().hipStreamCreateWithFlags(C.byref(h), 1), "hipStreamCreateWithFlags")     # hipStreamNonBlocking
            s = streams[dev] = h.value
        return s

    class _on:
        """Make ``device`` current for the duration of an allocation / copy when it is not already (a caller
        passing another device must not silently get memory and a stream on the current one)."""

        def __init__(self, device):
            index = getattr(device, "index", None)
            self.ctx = device_ctx(index) if index is not None and index != current_device() else None

        def __enter__(self):
            if self.ctx is not None:
                self.ctx.__enter__()

        def __exit__(self, *exc):
            if self.ctx is not None:
                self.ctx.__exit__(*exc)
            return False

    def synchronize():
        _ck(_rt().hipDeviceSynchronize(), "hipDeviceSynchronize")

    class Event:
        def __init__(self, enable_timing=False):
            self.h = C.c_void_p()
            _ck(_rt().hipEventCreateWithFlags(C.byref(self.h), 0 if enable_timing else 2), "hipEventCreateWithFlags")

        def record(self):
            _ck(_rt().hipEventRecord(self.h, C.c_void_p(stream_ptr())), "hipEventRecord")

        def synchronize(self):
            _ck(_rt().hipEventSynchronize(self.h), "hipEventSynchronize")

        def elapsed_time(self, other):
            ms = C.c_float(0)
            _ck(_rt().hipEventElapsedTime(C.byref(ms), self.h, other.h), "hipEventElapsedTime")
            return ms.value

        def __del__(self):
            try:
                if self.h:
                    _rt().hipEventDestroy(self.h)
            except Exception:
                pass

    class Graph:
        """A launch sequence captured once into a hipGraph (relaxed mode, on this thread's stream) and replayed there."""

        def __init__(self, enqueue):
            rt, s = _rt(), C.c_void_p(stream_ptr())
            graph, self.exe = C.c_void_p(), C.c_void_p()
            with _capture_lock:
                _ck(rt.hipStreamBeginCapture(s, 2), "hipStreamBeginCapture")          # hipStreamCaptureModeRelaxed
                try:
                    enqueue()
                except BaseException:
                    rt.hipStreamEndCapture(s, C.byref(graph))                         # leave capture mode, drop what was recorded
                    if graph:
                        rt.hipGraphDestroy(graph)
                    raise
                _ck(rt.hipStreamEndCapture(s, C.byref(graph)), "hipStreamEndCapture")
            try:
                _ck(rt.hipGraphInstantiate(C.byref(self.exe), graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")
            finally:
                rt.hipGraphDestroy(graph)

        def replay(self):
            _ck(_rt().hipGraphLaunch(self.exe, C.c_void_p(stream_ptr())), "hipGraphLaunch")

        def destroy(self):
            """Destroy the executable graph NOW: graphs that captured a communicator's operations must be gone before
            the communicator is (dist.HostComm.close)."""
            if self.exe:
                exe, self.exe = self.exe, C.c_void_p()
                _ck(_rt().hipGraphExecDestroy(exe), "hipGraphExecDestroy")

        def __del__(self):
            try:
                self.destroy()
            except Exception:
                pass

    class _Allocation:
        """Owns one hipMalloc / hipHostMalloc block; freed with the last array that views it."""

        def __init__(self, nbytes, pinned=False):
            self.ptr, self.pinned = C.c_void_p(), pinned
            call = _rt().hipHostMalloc if pinned else _rt().hipMalloc
            args = (C.byref(self.ptr), C.c_size_t(max(int(nbytes), 16))) + ((C.c_uint(0),) if pinned else ())
            _ck(call(*args), "hipHostMalloc" if pinned else "hipMalloc")

        def __del__(self):
            try:
                if self.ptr:
                    (_rt().hipHostFree if self.pinned else _rt().hipFree)(self.ptr)
            except Exception:
                pass

    class _Host:
        def __init__(self, a):
            self.a = a

        def numpy(self):
            return self.a

    def _host_array(src):
        """ndarray behind a copy source that lives on the host (ndarray, ``host(...)``, pinned array), else None."""
        if isinstance(src, np.ndarray):
            return src, False
        if isinstance(src, _Host):
            return src.a, False
        if isinstance(src, PinnedArray):
            return src.a, True
        return None, False

    class HipArray:
        """1-D contiguous or 2-D row-major (contiguous rows, any row pitch) view of device memory."""

        is_cuda = True

        def __init__(self, alloc, ptr, shape, pitch, dtype, device):
            self._alloc, self._ptr, self.shape, self._pitch = alloc, int(ptr), tuple(int(s) for s in shape), int(pitch)
            self.dtype, self.device = np.dtype(dtype), device          # pitch: elements between rows (2-D only)

        # -- facts
        def data_ptr(self):
            return self._ptr

        def numel(self):
            return int(np.prod(self.shape)) if self.shape else 1

        def is_contiguous(self):
            return len(self.shape) < 2 or self._pitch == self.shape[1] or self.shape[0] <= 1

        @property
        def ndim(self):
            return len(self.shape)

        # -- views
        def _row(self, i):
            n = self.shape[0]
            i = i + n if i < 0 else i
            if not 0 <= i < n:
                raise IndexError(i)
            return i

        @staticmethod
        def _span(sl, n):
            start, stop, step = sl.indices(n)
            if step != 1:
                raise NotImplementedError("HipArray slices have step 1")
            return start, max(stop, start)

        def __getitem__(self, key):
            item = self.dtype.itemsize
            if len(self.shape) == 1:
                if isinstance(key, slice):
                    a, b = self._span(key, self.shape[0])
                    return HipArray(self._alloc, self._ptr + a * item, (b - a,), 0, self.dtype, self.device)
                return HipArray(self._alloc, self._ptr + self._row(int(key)) * item, (), 0, self.dtype, self.device)
            rows, cols = (key if isinstance(key, tuple) else (key, slice(None)))
            c0, c1 = self._span(cols, self.shape[1])
            if isinstance(rows, slice):
                r0, r1 = self._span(rows, self.shape[0])
                return HipArray(self._alloc, self._ptr + (r0 * self._pitch + c0) * item, (r1 - r0, c1 - c0), self._pitch,
                                self.dtype, self.device)
            r = self._row(int(rows))
            return HipArray(self._alloc, self._ptr + (r * self._pitch + c0) * item, (c1 - c0,), 0, self.dtype, self.device)

        def view(self, dtype):
            dtype = np.dtype(dtype)
            if len(self.shape) != 1:
                raise NotImplementedError("HipArray.view: 1-D arrays only")
            nbytes = self.shape[0] * self.dtype.itemsize
            if nbytes % dtype.itemsize:
                raise ValueError("view: size is not a multiple of the new item size")
            return HipArray(self._alloc, self._ptr, (nbytes // dtype.itemsize,), 0, dtype, self.device)

        # -- data movement (all on this thread's stream)
        def _geometry(self):
            """(rows, row bytes, pitch bytes) of the region this view covers."""
            item = self.dtype.itemsize
            if len(self.shape) == 2:
                return self.shape[0], self.shape[1] * item, self._pitch * item
            return 1, self.numel() * item, self.numel() * item

        def _copy(self, dst_ptr, dst_pitch, src_ptr, src_pitch, rows, width, kind):
            s = C.c_void_p(stream_ptr())
            if rows == 1 or (dst_pitch == width and src_pitch == width):
                _ck(_rt().hipMemcpyAsync(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), C.c_size_t(rows * width), kind, s), "hipMemcpyAsync")
            else:
                _ck(_rt().hipMemcpy2DAsync(C.c_void_p(dst_ptr), C.c_size_t(dst_pitch), C.c_void_p(src_ptr), C.c_size_t(src_pitch),
                                           C.c_size_t(width), C.c_size_t(rows), kind, s), "hipMemcpy2DAsync")

        def copy_(self, src, non_blocking=False):
            rows, width, pitch = self._geometry()
            if isinstance(src, HipArray):
                if src.dtype != self.dtype or src.numel() != self.numel():
                    raise ValueError("copy_: shape / dtype mismatch")
                srows, swidth, spitch = src._geometry()
                if (srows, swidth) != (rows, width):
                    if not (src.is_contiguous() and self.is_contiguous()):
                        raise ValueError("copy_: incompatible strided shapes")
                    rows, width, pitch, spitch = 1, rows * width, rows * width, rows * width
                self._copy(self._ptr, pitch, src._ptr, spitch, rows, width, _D2D)
                return self
            a, pinned = _host_array(src)
            if a is None:
                raise TypeError(f"copy_ from {type(src).__name__}")
            a = np.ascontiguousarray(a)
            if a.dtype.itemsize * a.size != rows * width:
                raise ValueError("copy_: size mismatch")
            self._copy(self._ptr, pitch, a.ctypes.data, width, rows, width, _H2D)
            if not (pinned and non_blocking):        # a pageable source may be freed or rewritten as soon as we return
                _ck(_rt().hipStreamSynchronize(C.c_void_p(stream_ptr())), "hipStreamSynchronize")
            return self

        def zero_(self):
            rows, width, pitch = self._geometry()
            s = C.c_void_p(stream_ptr())
            if rows == 1 or pitch == width:
                _ck(_rt().hipMemsetAsync(C.c_void_p(self._ptr), 0, C.c_size_t(rows * width), s), "hipMemsetAsync")
            else:
                _ck(_rt().hipMemset2DAsync(C.c_void_p(self._ptr), C.c_size_t(pitch), 0, C.c_size_t(width), C.c_size_t(rows), s),
                    "hipMemset2DAsync")
            return self

        def cpu(self):
            rows, width, pitch = self._geometry()
            out = np.empty(self.shape, self.dtype)
            self._copy(out.ctypes.data, width, self._ptr, pitch, rows, width, _D2H)
            _ck(_rt().hipStreamSynchronize(C.c_void_p(stream_ptr())), "hipStreamSynchronize")
            return _Host(out)

        def item(self):
            assert self.numel() == 1
            return self.cpu().numpy().reshape(-1)[0].item()

    class PinnedArray:
        """Page-locked host memory (hipHostMalloc) with the few tensor methods the staging code uses."""

        is_cuda = False

        def __init__(self, alloc, a):
            self._alloc, self.a = alloc, a
            self.shape, self.dtype = a.shape, a.dtype

        def numpy(self):
            return self.a

        def numel(self):
            return self.a.size

        def __getitem__(self, key):
            return PinnedArray(self._alloc, self.a[key])

        def copy_(self, src, non_blocking=False):
            """device -> this pinned buffer, asynchronous on the thread's stream when ``non_blocking``."""
            assert isinstance(src, HipArray) and self.a.flags.c_contiguous
            rows, width, pitch = src._geometry()
            assert rows * width == self.a.nbytes
            src._copy(self.a.ctypes.data, width, src._ptr, pitch, rows, width, _D2H)
            if not non_blocking:
                _ck(_rt().hipStreamSynchronize(C.c_void_p(stream_ptr())), "hipStreamSynchronize")
            return self

    def _alloc_array(shape, dtype, device, zero):
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        device = as_device(device)
        switched = _on(device)                       # (decided BEFORE entering: inside, the target device is current)
        with switched:
            alloc = _Allocation(nbytes)
            arr = HipArray(alloc, alloc.ptr.value, shape, shape[1] if len(shape) == 2 else 0, dtype, device)
            if zero and nbytes:
                _ck(_rt().hipMemsetAsync(alloc.ptr, 0, C.c_size_t(nbytes), C.c_void_p(stream_ptr())), "hipMemsetAsync")
                if switched.ctx is not None:         # zeroed on the OTHER device's stream: done before that stream is left
                    _ck(_rt().hipStreamSynchronize(C.c_void_p(stream_ptr())), "hipStreamSynchronize")
        return arr

    def zeros(shape, dtype, device):
        return _alloc_array(shape, dtype, device, True)

    def empty(shape, dtype, device):
        return _alloc_array(shape, dtype, device, False)

    def upload(a, device):
        a = np.ascontiguousarray(a)
        out = _alloc_array(a.shape if a.ndim else (1,), a.dtype, device, False)
        if a.size:
            with _on(out.device):
                out.copy_(a.reshape(out.shape))      # (a pageable source: copy_ waits for the copy before it returns)
        return out

    def host(a):
        return _Host(np.ascontiguousarray(a))

    def pinned_empty(shape, dtype):
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        alloc = _Allocation(nbytes, pinned=True)
        buf = (C.c_char * max(nbytes, 1)).from_address(alloc.ptr.value)
        return PinnedArray(alloc, np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape))
