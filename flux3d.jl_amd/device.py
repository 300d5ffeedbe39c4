"""Device memory / stream plumbing on top of the C ABI (no torch needed).

Mirrors what the reference gets from CUDA.jl + Flux's functor walkers: ``gpu(x)`` / ``cpu(x)``
(src/Flux3D.jl:16-18, src/rep/pcloud.jl:57, src/rep/mesh.jl:189-190) and ``CuArray``.  A
:class:`DeviceArray` is the ``S`` storage type of ``TriMesh{T,R,S}`` / ``PointCloud``: a typed,
Julia-shaped (column-major) view of a device allocation.  Interop: anything exposing
``data_ptr()`` (a torch CUDA tensor) can be wrapped without a copy via :meth:`DeviceArray.wrap`.
"""
import ctypes as C

import numpy as np

from . import _lib


def device_count():
    n = C.c_int32(0)
    try:
        _lib.call("fx3d_device_count", C.byref(n))
    except _lib.Flux3DHipError:
        return 0
    return n.value


def functional():
    """`CUDA.functional()` analogue feeding the `use_cuda`-style flag (src/Flux3D.jl:52-61)."""
    return device_count() > 0


def set_device(dev):
    _lib.call("fx3d_set_device", int(dev))


def device_name(dev=0):
    buf = C.create_string_buffer(256)
    _lib.call("fx3d_device_name", int(dev), buf, 256)
    return buf.value.decode()


def device_identity(dev=0):
    """(pci_bus_id, uuid_hex) of device ``dev``: which physical GPU a rank sits on (bench.py ``comm.ranks``)."""
    buf = C.create_string_buffer(64)
    uuid = (C.c_uint8 * 16)()
    _lib.call("fx3d_device_identity", int(dev), buf, 64, uuid)
    return buf.value.decode(), bytes(uuid).hex()


def synchronize():
    _lib.call("fx3d_device_sync")


class Stream:
    """A HIP stream owned by the library (``None``/0 handle = the default stream)."""

    def __init__(self, handle=None, owned=False):
        self.handle = handle
        self._owned = owned

    @classmethod
    def create(cls):
        h = C.c_void_p()
        _lib.call("fx3d_stream_create", C.byref(h))
        _live_streams.add(h.value)
        return cls(h.value, owned=True)

    def synchronize(self):
        _lib.call("fx3d_stream_sync", self.handle)

    def wait_event(self, ev):
        """Work queued on this stream after the call starts only when ``ev`` (recorded on another stream) has completed;
        inside a Graph capture this is how a second stream joins the recording (fork) and returns to it (join)."""
        _lib.call("fx3d_stream_wait_event", self.handle, ev.handle)

    def __del__(self):
        if getattr(self, "_owned", False) and self.handle:
            try:
                _live_streams.discard(self.handle)
                _lib.load().fx3d_stream_destroy(self.handle)  # (hipStreamDestroy lets queued work finish)
            except Exception:
                pass
            self.handle = None


_live_streams = set()  # handles created here and not yet destroyed (the pool only synchronises these)


def _sync_handle(h):
    """Synchronise stream handle h (0 / None = the default stream) if it still exists."""
    if not h:
        _lib.load().fx3d_stream_sync(None)
    elif h in _live_streams:
        _lib.load().fx3d_stream_sync(h)
    else:  # destroyed, or a foreign handle: fall back to the whole device
        _lib.load().fx3d_device_sync()


DEFAULT_STREAM = Stream(None)
_current = [DEFAULT_STREAM]


def current_stream():
    return _current[-1]


class stream:
    """``with stream(s):`` routes every op of this package to ``s``."""

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        _current.append(self.s)
        return self.s

    def __exit__(self, *a):
        _current.pop()


class Event:
    """``timing=False``: an ordering-only event (fx3d_event_create_sync: no timestamps, no system-scope fence)."""

    def __init__(self, timing=True):
        h = C.c_void_p()
        _lib.call("fx3d_event_create" if timing else "fx3d_event_create_sync", C.byref(h))
        self.handle = h.value

    def record(self, s=None):
        _lib.call("fx3d_event_record", self.handle, (s or current_stream()).handle)

    def synchronize(self):
        _lib.call("fx3d_event_sync", self.handle)

    def elapsed_ms(self, later):
        ms = C.c_float(0)
        _lib.call("fx3d_event_elapsed_ms", self.handle, later.handle, C.byref(ms))
        return ms.value

    def __del__(self):
        if getattr(self, "handle", None):
            try:
                _lib.load().fx3d_event_destroy(self.handle)
            except Exception:
                pass
            self.handle = None


class _Pool:
    """Caching device allocator (what CUDA.jl's pool / torch's caching allocator give the reference's
    GPU path): hipMalloc/hipFree synchronise the device and cost tens of microseconds, so freed
    blocks are kept in power-of-two size classes and reused.  ``empty_cache()`` returns them.

    Reuse is STREAM-ORDERED: a block is cached under the stream that was current when it was allocated, and handed
    out again only to an allocation made under that same stream -- work enqueued there later runs after whatever
    still reads or writes the block.  A block released while a different stream is current (it may have been used on
    either) goes back only after both streams have drained; an allocation that finds nothing under its own stream may
    take a block cached under another one after synchronising that stream.  (the pool used to ignore
    streams, so a temporary released after an async launch on stream A could be handed at once to work on stream B.)"""

    def __init__(self, limit_bytes=4 << 30):
        self.free = {}  # (stream handle or 0, size class) -> [ptr]
        self.cached = 0
        self.limit = limit_bytes
        self.steals = 0  # allocations served from another stream's cache (after synchronising it)
        self.limbo = []  # blocks released across streams during a capture (no synchronisation there): freed by empty()

    @staticmethod
    def _cls(nbytes):
        n = 256
        while n < nbytes:
            n <<= 1
        return n

    def alloc(self, nbytes):
        c = self._cls(max(int(nbytes), 1))
        sh = current_stream().handle or 0
        lst = self.free.get((sh, c))
        if lst:
            self.cached -= c
            return lst.pop(), c, sh
        for (osh, oc), olst in self.free.items():
            if _capturing:
                break  # a capturing stream cannot be synchronised: no block crosses streams inside a capture
            if oc == c and olst and osh != sh:
                _sync_handle(osh)  # its pending work may still touch the block
                self.cached -= c
                self.steals += 1
                return olst.pop(), c, sh
        p = C.c_void_p()
        try:
            _lib.call("fx3d_malloc", C.byref(p), c)
        except _lib.Flux3DHipError:
            self.empty()
            _lib.call("fx3d_malloc", C.byref(p), c)
        return p.value, c, sh

    def release(self, ptr, c, sh=0):
        cur = current_stream().handle or 0
        if cur != sh and _capturing:  # two-branch capture: nothing may synchronise; the graph's pool keeps the block
            self.limbo.append(ptr)
            return
        if cur != sh:  # possibly used on both: neither may still be running on it when it is handed out again
            _sync_handle(cur)
            _sync_handle(sh)
        if self.cached + c > self.limit:
            if cur == sh:
                _sync_handle(sh)  # hipFree of a block with work in flight
            _lib.load().fx3d_free(ptr)
            return
        self.free.setdefault((sh, c), []).append(ptr)
        self.cached += c

    def empty(self):
        lib = _lib.load()
        lib.fx3d_device_sync()
        for lst in list(self.free.values()) + [self.limbo]:
            for ptr in lst:
                lib.fx3d_free(ptr)
        self.free.clear()
        self.limbo = []
        self.cached = 0


_pools = {}
_pool_override = []  # innermost private pool of an active Graph capture
_capturing = []      # the Graph objects being captured (innermost last)


def _pool():
    if _pool_override:
        return _pool_override[-1]
    d = C.c_int32(0)
    _lib.load().fx3d_get_device(C.byref(d))
    pl = _pools.get(d.value)
    if pl is None:
        pl = _pools[d.value] = _Pool()
    return pl


class Graph:
    """A captured sequence of this package's device ops (hipGraph), replayed with one launch.

        g = Graph()
        with g.capture(stream):          # everything enqueued on `stream` inside the block is recorded, not run
            step()
        g.launch()                       # runs the recorded sequence on `stream`

    Arrays allocated inside the block come from a pool private to the graph, so the addresses baked into the
    recording are never handed to anybody else while the graph is alive.  Run the block once eagerly on the same
    stream before capturing (grow-only workspaces, per-kernel attribute caches, optimiser state), and keep host
    round trips (``sync=True`` paths, ``to_host``) out of it."""

    def __init__(self):
        self.handle = None
        self.stream = None
        self._pool = _Pool(limit_bytes=1 << 62)
        self._ws = {}  # (stream, tag) -> Workspace used by the recorded ops

    def capture(self, s):
        return _GraphCapture(self, s)

    def launch(self, s=None):
        _lib.call("fx3d_graph_launch", self.handle, (s or self.stream).handle)

    def __del__(self):
        if getattr(self, "handle", None):
            try:
                _lib.load().fx3d_graph_destroy(self.handle)
                self._ws.clear()
                self._pool.empty()
            except Exception:
                pass
            self.handle = None


class _GraphCapture:
    def __init__(self, g, s):
        if s is None or not s.handle:
            raise ValueError("Graph.capture needs a created stream (Stream.create()), not the default stream")
        self.g, self.s = g, s

    def __enter__(self):
        _current.append(self.s)
        _pool_override.append(self.g._pool)
        _capturing.append(self.g)
        _lib.call("fx3d_graph_begin_capture", self.s.handle)
        return self.g

    def __exit__(self, exc_type, exc, tb):
        h = C.c_void_p()
        try:
            _lib.call("fx3d_graph_end_capture", self.s.handle, C.byref(h))
            self.g.handle, self.g.stream = h.value, self.s
        except Exception:
            if exc_type is None:
                raise
        finally:
            _capturing.pop()
            _pool_override.pop()
            _current.pop()
        return False


def empty_cache():
    """Return every cached block to the driver (cf. CUDA.reclaim / torch.cuda.empty_cache)."""
    for pl in _pools.values():
        pl.empty()


class DeviceArray:
    """Device buffer with a Julia-style (column-major) shape.  ``to_host()`` returns an
    F-contiguous numpy array of the same shape: byte-identical to the Julia ``Array``."""

    __slots__ = ("ptr", "shape", "dtype", "_owned", "_keep", "_pool", "_cls", "_stream")

    def __init__(self, ptr, shape, dtype, owned=False, keep=None, pool=None, cls=0, stream=0):
        self.ptr = ptr
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._owned = owned
        self._keep = keep
        self._pool = pool
        self._cls = cls
        self._stream = stream  # handle of the stream current at allocation (the pool's reuse key)

    # -- construction ---------------------------------------------------------------------
    @classmethod
    def empty(cls, shape, dtype=np.float32):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        pl = _pool()
        ptr, c, sh = pl.alloc(nbytes)
        return cls(ptr, shape, dtype, owned=True, pool=pl, cls=c, stream=sh)

    @classmethod
    def zeros(cls, shape, dtype=np.float32):
        a = cls.empty(shape, dtype)
        _lib.call("fx3d_memset", a.ptr, 0, max(a.nbytes, 1), current_stream().handle)
        return a

    @classmethod
    def from_host(cls, arr, dtype=None):
        arr = np.asfortranarray(arr, dtype=dtype)
        a = cls.empty(arr.shape, arr.dtype)
        if arr.nbytes:
            _lib.call("fx3d_memcpy_h2d", a.ptr, arr.ctypes.data, arr.nbytes, current_stream().handle)
        return a

    @classmethod
    def wrap(cls, obj, shape=None, dtype=np.float32):
        """Zero-copy view of foreign device memory (e.g. a torch CUDA tensor).  The tensor's
        memory is interpreted as the column-major array ``shape`` (default: reversed dims)."""
        ptr = obj.data_ptr()
        if shape is None:
            shape = tuple(reversed(tuple(obj.shape)))
        return cls(ptr, shape, dtype, owned=False, keep=obj)

    # -- properties -----------------------------------------------------------------------
    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        assert int(np.prod(shape, dtype=np.int64)) == self.size
        return DeviceArray(self.ptr, shape, self.dtype, owned=False, keep=self)

    def slab(self, start, count):
        """Contiguous sub-range along the LAST (slowest) dimension -- e.g. a batch shard."""
        inner = int(np.prod(self.shape[:-1], dtype=np.int64))
        off = start * inner * self.dtype.itemsize
        return DeviceArray(self.ptr + off, self.shape[:-1] + (count,), self.dtype, owned=False, keep=self)

    # -- transfers ------------------------------------------------------------------------
    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype, order="F")
        if self.nbytes:
            _lib.call("fx3d_memcpy_d2h", out.ctypes.data, self.ptr, self.nbytes, current_stream().handle)
        return out

    def copy_(self, host):
        host = np.asfortranarray(host, dtype=self.dtype)
        assert host.size == self.size
        _lib.call("fx3d_memcpy_h2d", self.ptr, host.ctypes.data, self.nbytes, current_stream().handle)
        return self

    def clone(self):
        out = DeviceArray.empty(self.shape, self.dtype)
        _lib.call("fx3d_memcpy_d2d", out.ptr, self.ptr, self.nbytes, current_stream().handle)
        return out

    def item(self):
        return self.to_host().reshape(-1)[0]

    def __del__(self):
        if getattr(self, "_owned", False) and self.ptr:
            try:
                # stream-ordered reuse: the block goes back to the cache of the stream it was allocated under
                # (see _Pool); a recycled block is only touched by work enqueued later on that stream
                if self._pool is not None:
                    self._pool.release(self.ptr, self._cls, self._stream)
                else:
                    _lib.load().fx3d_free(self.ptr)
            except Exception:
                pass
            self.ptr = None

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, ptr=0x{(self.ptr or 0):x})"


def is_device(x):
    return isinstance(x, DeviceArray)


def gpu(x):
    """`gpu(x)` (src/Flux3D.jl:16-18): move an array / PointCloud / TriMesh to the device."""
    if hasattr(x, "_to_device"):
        return x._to_device()
    if is_device(x):
        return x
    return DeviceArray.from_host(np.asarray(x))


def cpu(x):
    """`cpu(x)`: bring an array / PointCloud / TriMesh back to host numpy arrays."""
    if hasattr(x, "_to_host"):
        return x._to_host()
    if is_device(x):
        return x.to_host()
    return x


class Workspace:
    """Grow-only scratch buffer handed to the ops that need one (caller-provided scratch is part
    of the C ABI contract).  One per stream in use; the default one serves the default stream."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes):
        if self.buf is None or self.buf.nbytes < nbytes:
            self.buf = DeviceArray.empty((max(int(nbytes), 4096),), np.uint8)
        return self.buf


_workspaces = {}


def workspace(nbytes, tag="default"):
    """The grow-only scratch of (current stream, tag).  Inside a Graph capture the scratch is private to the graph
    (allocated from its pool, kept alive by it): the recording bakes the address in, and a later, larger eager call
    on the same stream must not regrow -- and thereby release -- a buffer that graph replays still write."""
    key = (current_stream().handle, tag)
    table = _capturing[-1]._ws if _capturing else _workspaces
    ws = table.get(key)
    if ws is None:
        ws = table[key] = Workspace()
    return ws.get(nbytes)
