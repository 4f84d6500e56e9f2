"""CPU-only host logic added in round 6: the default (torch-free) allocator backend, the serialisation of hipGraph
captures, the order "graphs before communicator", the capture policy for sequences with collectives, and the hardened
TCP rendezvous.  The HIP runtime is a recording stub here (no GPU): what is checked is which calls are made, in which
order, on which device."""
import ctypes as C
import importlib.util
import os
import socket
import struct
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

from conftest import PKG_DIR, ROOT


# ---------------------------------------------------------------------------- the default backend
def test_default_backend_is_the_hip_runtime_and_needs_no_torch():
    """without AKS_HOST_ALLOC the package runs on the HIP runtime alone -- importing it (and its whole
    host layer) does not import torch, although torch IS installed here -- and pyproject.toml asks for numpy and scipy only,
    the reference's own dependencies (/root/reference/pyproject.toml:9-13), with torch as an optional extra."""
    env = {k: v for k, v in os.environ.items() if k not in ("AKS_HOST_ALLOC", "AKS_TEST_BACKEND")}
    code = ("import sys; sys.path.insert(0, %r); import arnoldi_amd; from arnoldi_amd import mem, engine, dist, harness, explicit_restarts; "
            "print(mem.BACKEND, 'torch' in sys.modules)" % PKG_DIR)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["hip", "False"], out.stdout
    assert importlib.util.find_spec("torch") is not None            # (it could have been imported: it was not)
    text = open(os.path.join(ROOT, "pyproject.toml")).read()
    deps = text[text.index("dependencies = ["):].split("]")[0]
    assert "numpy" in deps and "scipy" in deps and "torch" not in deps
    assert 'torch = ["torch"]' in text


# ---------------------------------------------------------------------------- a recording stand-in for libamdhip64
class StubRuntime:
    """The handful of HIP runtime entry points ``mem``'s HIP backend calls, recorded.  ``byref`` arguments are written
    through ``arg._obj``."""

    def __init__(self):
        self.log, self.device, self.lock = [], 0, threading.Lock()
        self._next = 0x1000

    def _handle(self):
        self._next += 0x1000
        return self._next

    def _rec(self, *entry):
        with self.lock:
            self.log.append(entry)

    def hipGetDevice(self, out):
        out._obj.value = self.device
        return 0

    def hipSetDevice(self, index):
        self.device = int(index)
        self._rec("set_device", int(index))
        return 0

    def hipGetDeviceCount(self, out):
        out._obj.value = 2
        return 0

    def hipMalloc(self, out, nbytes):
        out._obj.value = self._handle()
        self._rec("malloc", self.device)
        return 0

    def hipFree(self, ptr):
        self._rec("free",)
        return 0

    def hipStreamCreateWithFlags(self, out, flags):
        out._obj.value = self._handle() + self.device           # a stream belongs to the device it was made on
        self._rec("stream_create", self.device, out._obj.value)
        return 0

    def hipMemsetAsync(self, ptr, value, nbytes, stream):
        self._rec("memset", self.device, stream.value)
        return 0

    def hipStreamSynchronize(self, stream):
        self._rec("stream_sync", self.device, stream.value)
        return 0

    def hipStreamBeginCapture(self, stream, mode):
        self._rec("begin_capture", threading.current_thread().name)
        return 0

    def hipStreamEndCapture(self, stream, graph_out):
        graph_out._obj.value = self._handle()
        self._rec("end_capture", threading.current_thread().name)
        return 0

    def hipGraphInstantiate(self, exe_out, graph, a, b, c):
        exe_out._obj.value = self._handle()
        self._rec("instantiate", threading.current_thread().name)
        return 0

    def hipGraphDestroy(self, graph):
        return 0

    def hipGraphExecDestroy(self, exe):
        self._rec("exec_destroy",)
        return 0

    def hipGraphLaunch(self, exe, stream):
        self._rec("launch",)
        return 0

    def hipGetErrorString(self, status):
        return b"stub"


@pytest.fixture
def hip_mem(monkeypatch):
    """``arnoldi_amd/mem.py`` loaded a second time as its HIP backend (this process runs the torch one), on the stub."""
    monkeypatch.setenv("AKS_HOST_ALLOC", "hip")
    spec = importlib.util.spec_from_file_location("arnoldi_amd_mem_hip_for_tests", os.path.join(PKG_DIR, "arnoldi_amd", "mem.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.BACKEND == "hip"
    stub = StubRuntime()
    mod._rt.lib = stub
    return mod, stub


def test_zeros_on_another_device_waits_for_its_memset(hip_mem):
    """``zeros()`` for a device that is not current zeroes on THAT device's stream and must wait for the memset
    before it leaves the device again -- the check used to be evaluated after the switch (``_on(device)`` inside ``with
    _on(device)``: always "already current"), so the synchronize was dead code and later work on the caller's stream was
    not ordered behind the zeroing."""
    mem, rt = hip_mem
    rt.device = 0
    keep = [mem.zeros(1024, mem.f64, mem.Device(1))]
    names = [e[0] for e in rt.log]
    assert names.index("set_device") < names.index("malloc") < names.index("memset") < names.index("stream_sync")
    memset, sync = rt.log[names.index("memset")], rt.log[names.index("stream_sync")]
    assert memset[1] == 1 and sync[1] == 1 and memset[2] == sync[2]           # on device 1, the same (device-1) stream
    assert rt.log[-1] == ("set_device", 0) and rt.device == 0                 # and only then back to the caller's device
    # the current device: no switch, and no synchronisation either (the caller's stream orders the memset)
    rt.log.clear()
    keep.append(mem.zeros(1024, mem.f64, mem.Device(0)))
    names = [e[0] for e in rt.log]
    assert "set_device" not in names and "stream_sync" not in names and "memset" in names
    # empty(): nothing to wait for on either device
    rt.log.clear()
    keep.append(mem.empty(1024, mem.f64, mem.Device(1)))
    assert "stream_sync" not in [e[0] for e in rt.log]


def test_graph_captures_of_two_threads_never_overlap(hip_mem):
    """What fixed round 5's ``hipErrorStreamCaptureInvalidated`` ("operation failed due to a previous error during
    capture", gpurun_out/r05_suite_again.log: one thread's capture entry freed device memory -- torch's cache-emptying
    ``cuda.graph()`` entry -- while the other thread's capture was open): captures are serialised over a process's host
    threads.  Deterministically: thread A is held INSIDE its capture; thread B's capture must not begin before A's has
    ended.  (the cause pinned on the host, instead of 25 GPU runs hoping to see it again.)"""
    mem, rt = hip_mem
    inside, release = threading.Event(), threading.Event()

    def slow_enqueue():
        inside.set()
        assert release.wait(10)

    graphs = {}
    a = threading.Thread(target=lambda: graphs.__setitem__("A", mem.Graph(slow_enqueue)), name="A")
    b = threading.Thread(target=lambda: graphs.__setitem__("B", mem.Graph(lambda: None)), name="B")
    a.start()
    assert inside.wait(10)
    b.start()
    time.sleep(0.3)                                    # B has had every chance to begin
    assert [e for e in rt.log if e[0] == "begin_capture"] == [("begin_capture", "A")]
    release.set()
    a.join(10)
    b.join(10)
    order = [e for e in rt.log if e[0] in ("begin_capture", "end_capture")]
    assert order == [("begin_capture", "A"), ("end_capture", "A"), ("begin_capture", "B"), ("end_capture", "B")]
    # destroy() is immediate and idempotent (the communicator's close() relies on it)
    graphs["A"].destroy()
    graphs["A"].destroy()
    assert [e[0] for e in rt.log].count("exec_destroy") == 1


def test_the_torch_graph_entry_makes_no_device_wide_call(monkeypatch):
    """The other half of the same fix, on the torch backend: ``mem.Graph`` spells ``torch.cuda.graph()`` out WITHOUT its
    entry's ``torch.cuda.synchronize()`` / ``empty_cache()`` (the ``hipFree`` of an emptied cache is what invalidated the
    other thread's capture)."""
    import torch

    from arnoldi_amd import mem

    if mem.BACKEND != "torch":
        pytest.skip("torch backend only")
    calls = []

    class FakeStream:
        def wait_stream(self, other):
            calls.append("wait_stream")

    class FakeGraph:
        def capture_begin(self, **kw):
            calls.append(("capture_begin", kw.get("capture_error_mode")))

        def capture_end(self):
            calls.append("capture_end")

        def replay(self):
            calls.append("replay")

        def reset(self):
            calls.append("reset")

    class FakeCtx:
        def __init__(self, s):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    monkeypatch.setattr(torch.cuda, "CUDAGraph", FakeGraph)
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", FakeCtx)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: calls.append("DEVICE-WIDE synchronize"))
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: calls.append("DEVICE-WIDE empty_cache"))
    g = mem.Graph(lambda: calls.append("enqueue"))
    assert calls == ["wait_stream", ("capture_begin", "relaxed"), "enqueue", "capture_end", "wait_stream"], calls
    g.replay()
    g.destroy()
    g.destroy()
    assert calls[-2:] == ["replay", "reset"]


# ---------------------------------------------------------------------------- graphs before the communicator
class StubLibrary:
    """The communicator entry points the registry uses, with the library's own rule: destroy refuses while it counts a graph."""

    def __init__(self):
        self.count, self.log = 0, []

    def aks_comm_graph_retain(self, handle):
        self.count += 1
        self.log.append("retain")
        return self.count

    def aks_comm_graph_release(self, handle):
        self.count -= 1
        self.log.append("release")
        return self.count

    def aks_comm_destroy(self, handle):
        self.log.append("destroy" if self.count == 0 else "destroy refused")
        return 0 if self.count == 0 else -1

    def aks_last_error(self):
        return b"1 hipGraph(s) that captured operations of this communicator are still alive"


def test_close_drops_the_graphs_before_the_communicator(monkeypatch):
    """``ncclCommDestroy`` never returns while a hipGraph holds a captured send / recv of the communicator
    (profiles/r05_capture_crash.txt section 4): ``Comm.close()`` / ``HostComm.close()`` destroy every registered context's
    graphs FIRST; a graph that only the library's count knows of turns into ``HipLibraryError`` -- the communicator is
    left intact -- instead of a hang."""
    from arnoldi_amd import _hip
    from arnoldi_amd.dist import HostComm
    from arnoldi_amd.engine import ArnoldiContext

    lib = StubLibrary()
    monkeypatch.setattr(_hip, "load", lambda: lib)
    comm = HostComm(rank=0, size=1)
    comm._native = C.c_void_p(0x1234)

    class G:
        def __init__(self):
            self.alive = True

        def destroy(self):
            self.alive = False
            lib.log.append("graph destroyed")

    ctx = ArnoldiContext.__new__(ArnoldiContext)
    ctx.comm, ctx._graphs, ctx._graphs_on_comm = comm, {}, 0
    for key in ("a", "b"):
        ctx._graphs[key] = G()
        comm.adopt_graph_owner(ctx)
        ctx._graphs_on_comm += 1
    assert lib.count == 2
    # a graph nobody registered (only the library counts it): close() refuses loudly, communicator intact
    lib.aks_comm_graph_retain(comm._native)
    with pytest.raises(_hip.HipLibraryError, match="still alive"):
        comm._destroy_native()
    assert comm._native is not None and lib.log[-1] == "destroy refused" and not ctx._graphs and ctx._graphs_on_comm == 0
    assert lib.log.count("graph destroyed") == 2 and lib.count == 1
    lib.aks_comm_graph_release(comm._native)
    comm.close()
    assert comm._native is None and lib.log[-1] == "destroy"
    order = [e for e in lib.log if e in ("graph destroyed", "destroy")]
    assert order == ["graph destroyed", "graph destroyed", "destroy"]


@pytest.mark.parametrize("mode,exchange,version,want", [
    ("0", False, 70200000, False), ("1", False, 70000000, True), ("1", True, 70200000, False),
    ("exchange", True, 70200000, True), ("exchange", True, 70051831, False), ("exchange", False, 70051831, True),
    (None, False, 70200000, False)])
def test_capture_policy_for_sequences_with_collectives(monkeypatch, mode, exchange, version, want):
    """AKS_GRAPH_COMM: never by default; ``1`` = reductions only; ``exchange`` = also the ghost exchange, but ONLY on a
    HIP runtime >= 7.2 -- on the 7.0.51831 a torch wheel bundles the end-of-capture walk recurses without bound
    (profiles/r05_capture_crash.txt), so there the switch is ignored and the sequence stays eager."""
    from arnoldi_amd import _hip
    from arnoldi_amd.engine import ArnoldiContext

    if mode is None:
        monkeypatch.delenv("AKS_GRAPH_COMM", raising=False)
    else:
        monkeypatch.setenv("AKS_GRAPH_COMM", mode)
    monkeypatch.setattr(_hip, "runtime_versions", lambda: {"hip_runtime": version, "hip_driver": version, "rccl": None})
    ctx = ArnoldiContext.__new__(ArnoldiContext)
    ctx.op = type("Op", (), {"any_exchange": exchange})()
    assert ctx._comm_capturable() is want


# ---------------------------------------------------------------------------- the rendezvous
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _hub_pair(monkeypatch, port, body1, timeout=10.0):
    """Rank 0's hub in this thread, rank 1's in another; returns (hub0, result of body1(hub1))."""
    from arnoldi_amd.dist import _Hub

    box = {}

    def rank1():
        try:
            h = _Hub(1, 2, "127.0.0.1", port, timeout)
            box["out"] = body1(h)
            box["hub"] = h
        except Exception as e:                      # noqa: BLE001
            box["err"] = e

    t = threading.Thread(target=rank1)
    t.start()
    return box, t


def test_rendezvous_drops_a_stranger_and_still_admits_the_job(monkeypatch):
    """the listener used to trust whoever connected and claimed a rank.  A connection that does not open with
    the job's token (AKS_COMM_TOKEN, or a digest of the launcher's coordinates) is dropped -- its socket closed -- and does
    not take the rank slot: the real rank 1 still gets in."""
    from arnoldi_amd.dist import _Hub

    monkeypatch.setenv("AKS_COMM_TOKEN", "job-1234")
    port = _free_port()

    def stranger():
        for _ in range(200):
            try:
                s = socket.create_connection(("127.0.0.1", port), timeout=1)
                break
            except OSError:
                time.sleep(0.02)
        s.sendall(struct.pack("<q", 1) + b"x" * 32)              # claims rank 1 with the wrong token
        s.settimeout(5)
        try:
            closed = s.recv(1) == b""
        except OSError:
            closed = True
        box["stranger_closed"] = closed
        s.close()
        go.set()

    box, go = {}, threading.Event()
    threading.Thread(target=stranger).start()

    def real_rank():
        assert go.wait(10)                                       # after the stranger has been turned away
        h = _Hub(1, 2, "127.0.0.1", port, 10.0)
        box["got"] = h.gather(b"one")
        h.close()

    t = threading.Thread(target=real_rank)
    t.start()
    hub0 = _Hub(0, 2, "127.0.0.1", port, 15.0)
    assert hub0.gather(b"zero") == [b"zero", b"one"]
    t.join(10)
    hub0.close()
    assert box["stranger_closed"] and box["got"] == [b"zero", b"one"]


def test_rendezvous_messages_are_capped_and_read_in_chunks(monkeypatch):
    """A peer that announces more than AKS_COMM_MAX_MSG bytes is refused BEFORE anything is allocated for it; a legitimate
    message larger than one chunk arrives whole (memory grows with what has arrived, not with what was announced)."""
    from arnoldi_amd import dist

    monkeypatch.setenv("AKS_COMM_TOKEN", "job-5678")
    monkeypatch.setenv("AKS_COMM_MAX_MSG", str(1 << 20))
    monkeypatch.setattr(dist._Hub, "CHUNK", 4096)
    port = _free_port()
    payload = np.random.default_rng(0).integers(0, 256, 300_000, dtype=np.uint8).tobytes()     # 74 chunks

    def body1(h):
        first = h.gather(payload)
        # now a malformed announcement, by hand: one part of 2**40 bytes
        h.peers[0].sendall(struct.pack("<qq", 1, 1 << 40))
        return first

    box, t = _hub_pair(monkeypatch, port, body1)
    hub0 = dist._Hub(0, 2, "127.0.0.1", port, 10.0)
    assert hub0.gather(b"abc") == [b"abc", payload]
    with pytest.raises(RuntimeError, match="more than AKS_COMM_MAX_MSG"):
        hub0._recv_blobs(hub0.peers[1])
    t.join(10)
    assert "err" not in box and box["out"] == [b"abc", payload]
    # the sender's side of the same limit
    with pytest.raises(RuntimeError, match="exceeds AKS_COMM_MAX_MSG"):
        hub0._send_blobs(hub0.peers[1], [b"\0" * ((1 << 20) + 1)])
    hub0.close()
    box["hub"].close()


def test_rendezvous_finds_its_port_when_the_first_one_is_taken(monkeypatch):
    """The default address is MASTER_PORT + 1 -- a port nobody reserved.  If some other service owns it, rank 0 listens on
    the next port of the span it can bind, and the other ranks find it by the handshake: the foreign listener does not
    answer the hello with the job's token, so they move on (a driver-started run must not die of a port collision)."""
    from arnoldi_amd.dist import _Hub

    monkeypatch.setenv("AKS_COMM_TOKEN", "job-span")
    foreign = socket.socket()
    foreign.bind(("127.0.0.1", 0))
    foreign.listen(4)
    port = foreign.getsockname()[1]

    def foreign_service():                          # accepts, says something else, hangs up
        foreign.settimeout(10)
        try:
            while True:
                c, _ = foreign.accept()
                c.sendall(b"HTTP/1.1 400 Bad Request\r\n\r\n")
                c.close()
        except OSError:
            pass

    threading.Thread(target=foreign_service, daemon=True).start()
    box = {}

    def rank1():
        h = _Hub(1, 2, "127.0.0.1", port, 10.0, span=4)
        box["port"], box["got"] = h.port, h.gather(b"one")
        h.close()

    t = threading.Thread(target=rank1)
    t.start()
    hub0 = _Hub(0, 2, "127.0.0.1", port, 10.0, span=4)
    assert hub0.port != port and port < hub0.port < port + 4
    assert hub0.gather(b"zero") == [b"zero", b"one"]
    t.join(10)
    hub0.close()
    foreign.close()
    assert box["port"] == hub0.port and box["got"] == [b"zero", b"one"]
    # an explicit address names ONE port: taken means an error that says what to do
    busy = socket.socket()
    busy.bind(("127.0.0.1", 0))
    busy.listen(1)
    with pytest.raises(RuntimeError, match="AKS_RENDEZVOUS=host:port"):
        _Hub(0, 2, "127.0.0.1", busy.getsockname()[1], 2.0)
    busy.close()


def test_a_device_that_is_not_current_is_refused(monkeypatch):
    """A solve launches on the CURRENT device's stream (one process per GPU).  Asking for another device used to allocate
    there and launch here; now it is refused with the remedy in the message (and the current / default device still passes)."""
    from arnoldi_amd import _hip, device as dev, mem

    inited = []
    monkeypatch.setattr(mem, "gpu_available", lambda: True)
    monkeypatch.setattr(mem, "current_device", lambda: 0)
    monkeypatch.setattr(mem, "as_device", lambda d: type("D", (), {"index": d})())
    monkeypatch.setattr(_hip, "device_init", lambda i: inited.append(i))
    with pytest.raises(_hip.HipLibraryError, match=r"make it current first \(arnoldi_amd.mem.set_device\(1\)"):
        dev._require_gpu(1)
    assert dev._require_gpu(0).index == 0 and dev._require_gpu(None).index is None and inited == [0, 0]
