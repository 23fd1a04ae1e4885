"""ctypes binding over libmodgpu.so -- every call goes through the C ABI of include/modgpu.h.

Nothing is computed in Python: if the library is missing or a call fails, ModGpuError is raised.

Two flavours of the library exist (modulate_amd/csrc/Makefile): the shipped libmodgpu.so, which is what every
call here uses by default, and libmodgpu_testing.so -- the same sources and device code plus the
modgpu_debug_* hooks of include/modgpu_testing.h.  `with testing_flavour():` routes the calls made inside
the block to the testing flavour (both can be loaded in one process); the debug_* functions refuse to run
outside such a block, because the shipped library has no such hooks.
"""
import contextlib
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KEY_PS3 = 0xC64EED30  # Modulate/Settings.h:19
KEY_PS4 = 0x90CFC0AB  # Modulate/Settings.h:20
MAGIC_PS3 = 0xC64EED30  # Modulate/Settings.h:16
MAGIC_PS4 = 0x6F303F55  # Modulate/Settings.h:17

# every symbol include/modgpu.h declares: (name, restype, argtypes)
_u64, _i32, _int, _vp = ctypes.c_uint64, ctypes.c_int32, ctypes.c_int, ctypes.c_void_p
EXPORTS = {
    "modgpu_abi_version": (_int, []),
    "modgpu_device_count": (_int, []),
    "modgpu_last_error": (ctypes.c_char_p, []),
    "modgpu_cycle_device": (_int, [_vp, _u64, _i32, _u64, _int, _vp]),
    "modgpu_cycle_host": (_int, [_vp, _u64, _i32, _u64, _int]),
    "modgpu_hdr_decrypt_host": (_int, [_vp, _u64, _int]),
    "modgpu_hdr_encrypt_host": (_int, [_vp, _u64, _int, _int]),
    "modgpu_cycle_parts_host": (_int, [ctypes.POINTER(_vp), ctypes.POINTER(_u64), _int, _i32, _int]),
    "modgpu_cycle_parts_device": (_int, [ctypes.POINTER(_vp), ctypes.POINTER(_u64), ctypes.POINTER(_int), _int, _i32]),
    "modgpu_cycle_host_split": (_int, [_vp, _u64, _i32, _u64, _int]),
    "modgpu_cycle_batch_device": (_int, [ctypes.POINTER(_vp), ctypes.POINTER(_u64), ctypes.POINTER(_u64), _int, _i32, _int, _vp]),
    "modgpu_cycle_file": (_int, [ctypes.c_char_p, ctypes.c_char_p, _i32, _u64, _int]),
    "modgpu_cycle_file_to_host": (_int, [ctypes.c_char_p, _u64, _vp, _u64, _i32, _u64, _int]),
    "modgpu_cycle_host_to_file": (_int, [_vp, _u64, ctypes.c_char_p, _i32, _u64, _int]),
    "modgpu_alloc": (_int, [ctypes.POINTER(_vp), _u64, _int]),
    "modgpu_free": (_int, [_vp, _int]),
    "modgpu_h2d": (_int, [_vp, _vp, _u64, _int]),
    "modgpu_d2h": (_int, [_vp, _vp, _u64, _int]),
    "modgpu_sync": (_int, [_int, _vp]),
    "modgpu_prepare": (_int, [_int]),
    "modgpu_state_at": (ctypes.c_uint32, [_i32, _u64]),
    "modgpu_jump_table": (_int, [_int, ctypes.POINTER(ctypes.c_uint32), _int]),
    "modgpu_cycle_scalar_host": (_int, [_vp, _u64, _i32, _u64]),
    "modgpu_cycle_auto_host": (_int, [_vp, _u64, _i32, _u64, _int]),
    "modgpu_host_alloc": (_int, [ctypes.POINTER(_vp), _u64]),
    "modgpu_host_free": (_int, [_vp]),
    "modgpu_host_is_pinned": (_int, [_vp, _u64]),
    "modgpu_host_register": (_int, [_vp, _u64]),
    "modgpu_host_unregister": (_int, [_vp]),
    "modgpu_path_stats": (_int, [_vp, _int]),
    "modgpu_gpu_required": (_int, []),
    "modgpu_min_gpu_bytes": (_u64, []),
    "modgpu_host_policy": (ctypes.c_char_p, []),
    "modgpu_host_policy_engine": (_int, [_u64, _int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "modgpu_host_loop_isa": (ctypes.c_char_p, []),
    "modgpu_host_alloc_near": (_int, [ctypes.POINTER(_vp), _u64, _int]),
    "modgpu_host_alloc_parts": (_int, [ctypes.POINTER(_vp), ctypes.POINTER(_u64), _int, _int]),
    "modgpu_device_numa_node": (_int, [_int]),
}


class PathStats(ctypes.Structure):
    """modgpu_path_stats_t (include/modgpu.h)."""
    _fields_ = [(k, _u64) for k in ("gpu_calls", "gpu_bytes", "gpu_launches", "scalar_calls", "scalar_bytes",
                                    "staged_bytes", "direct_bytes", "auto_fallbacks", "auto_small", "auto_policy_host", "midcall_rescues", "midcall_rescued_bytes")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class HostTraceEvent(ctypes.Structure):
    """modgpu_host_trace_event_t (include/modgpu_testing.h)."""
    _fields_ = [("t_ns", _u64), ("kind", _int), ("pipe", _int), ("chunk", _u64), ("bytes", _u64), ("tid", _int), ("reserved", _int)]


HOST_TRACE_KINDS = ("call_begin", "slots", "posted", "pipe_start", "fill_begin", "fill_end", "launched", "sync_begin", "sync_end",
                    "drain_end", "pipe_end", "call_end", "failed", "rescued", "ready")


class LaunchInfo(ctypes.Structure):
    """modgpu_launch_info_t (include/modgpu_testing.h)."""
    _fields_ = [("kernel", ctypes.c_char_p), ("variant", _int), ("grid", ctypes.c_uint32), ("block", ctypes.c_uint32),
                ("chunk_bytes", ctypes.c_uint32), ("bytes", _u64), ("main_groups", ctypes.c_uint32), ("source_hash", ctypes.c_char_p)]


# include/modgpu_testing.h, reporting group: in both flavours (measurement, not the drop-in boundary)
TESTING_EXPORTS = {
    "modgpu_time_cycle_device": (_int, [_vp, _u64, _i32, _u64, _int, _vp, _int, ctypes.POINTER(ctypes.c_float)]),
    "modgpu_last_launch": (_int, [ctypes.POINTER(LaunchInfo)]),
    "modgpu_kernel_source_hash": (ctypes.c_char_p, []),
    "modgpu_feed_kernel_source_hash": (ctypes.c_char_p, []),
    "modgpu_host_tunables": (None, [ctypes.POINTER(_u64)]),
    "modgpu_host_chunking": (None, [ctypes.POINTER(_u64)]),
    "modgpu_host_loop_info": (None, [ctypes.POINTER(_u64)]),
    "modgpu_host_trace": (None, [_int]),
    "modgpu_host_trace_read": (_int, [ctypes.POINTER(HostTraceEvent), _int]),
    "modgpu_host_pool_stats": (None, [ctypes.POINTER(_u64)]),
    "modgpu_cycle_scalar_host_isa": (_int, [_vp, _u64, _i32, _u64, ctypes.c_char_p]),
    "modgpu_queue_stats": (None, [ctypes.POINTER(_u64)]),
    "modgpu_host_alloc_on_node": (_int, [ctypes.POINTER(_vp), _u64, _int]),
    "modgpu_testing_hooks": (_int, []),
    "modgpu_numa_probe": (_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(_int), ctypes.POINTER(_int), _int]),
}
# include/modgpu_testing.h, modgpu_debug_* group: ONLY in libmodgpu_testing.so
DEBUG_EXPORTS = {
    "modgpu_debug_set_launch": (None, [_int, ctypes.c_uint32]),
    "modgpu_debug_set_pinned_mode": (None, [_int]),
    "modgpu_debug_set_staged_mode": (None, [_int]),
    "modgpu_debug_set_queue_ring": (None, [ctypes.c_uint32]),
    "modgpu_debug_set_helpers": (None, [_int]),
    "modgpu_debug_set_batch": (None, [_int]),
    "modgpu_debug_set_pcie_grid": (None, [ctypes.c_uint32]),
    "modgpu_debug_set_gpu_node": (None, [_int]),
    "modgpu_debug_set_host_tunable": (None, [_int, _u64]),
    "modgpu_debug_inject_failures": (None, [_int]),
    "modgpu_debug_inject_failure_at": (None, [ctypes.c_int64, _int]),
    "modgpu_debug_injection_armed": (_int, []),
    "modgpu_debug_hold_slots": (_int, [_int, _int]),
    "modgpu_debug_forbid_worker_threads": (None, [_int]),
}


class ModGpuError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"modgpu error {code}: {text}")
        self.code = code


FLAVOURS = {"shipped": "libmodgpu.so", "testing": "libmodgpu_testing.so"}


def lib_path(flavour="shipped"):
    # MODGPU_LIB: another build of the same sources stands in for both flavours (the sanitizer builds of
    # `make sanitize-lib`, which carry the hooks; tests/test_sanitizers.py)
    return os.environ.get("MODGPU_LIB") or os.path.join(_HERE, FLAVOURS[flavour])


_libs = {}
_active = "shipped"


def _load(flavour):
    if flavour not in _libs:
        path = lib_path(flavour)
        if not os.path.exists(path):
            raise ModGpuError(-1, f"{path} not built: run `make -C modulate_amd/csrc` (the Python package has no implementation of its own)")
        L = ctypes.CDLL(path)
        table = list(EXPORTS.items()) + list(TESTING_EXPORTS.items()) + (list(DEBUG_EXPORTS.items()) if flavour == "testing" else [])
        for name, (res, args) in table:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _libs[flavour] = L
    return _libs[flavour]


def lib():
    """The library calls go to: the shipped libmodgpu.so, or libmodgpu_testing.so inside `testing_flavour()`.
    Raises if it was not built (python __graft_entry__.py build)."""
    return _load(_active)


def active_flavour():
    return _active


@contextlib.contextmanager
def testing_flavour():
    """Calls made inside the block go to libmodgpu_testing.so (the flavour that has the modgpu_debug_* hooks).
    Buffers created inside keep using it after the block (they remember their library)."""
    global _active
    prev, _active = _active, "testing"
    try:
        yield _load("testing")
    finally:
        _active = prev


def use_testing_flavour():
    """For tools/: switch this process to libmodgpu_testing.so for good."""
    global _active
    _active = "testing"
    return _load("testing")


def _debug_lib():
    if _active != "testing":
        raise ModGpuError(-1, "modgpu_debug_* hooks exist only in libmodgpu_testing.so: call inside `with testing_flavour():`")
    return _load("testing")


def as_int32(key):
    key &= 0xFFFFFFFF
    return key - (1 << 32) if key & 0x80000000 else key


def _check(rc):
    if rc != 0:
        raise ModGpuError(rc, lib().modgpu_last_error().decode())


def _host_ptr(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"] and a.flags["WRITEABLE"]):
        raise TypeError("need a writable C-contiguous uint8 ndarray")
    return ctypes.c_void_p(a.ctypes.data)


def device_count():
    return lib().modgpu_device_count()


def cycle_host(buf, key, stream_off=0, device=-1):
    """In-place CEncryptionCycler::Cycle over a host ndarray, computed on the GPU."""
    _check(lib().modgpu_cycle_host(_host_ptr(buf), buf.size, as_int32(key), stream_off, device))
    return buf


def cycle_scalar_host(buf, key, stream_off=0):
    """The library's own host loop (never the GPU)."""
    _check(lib().modgpu_cycle_scalar_host(_host_ptr(buf), buf.size, as_int32(key), stream_off))
    return buf


def cycle_auto_host(buf, key, stream_off=0, device=-1):
    """What CEncryptionCycler::Cycle binds to: the GPU, the host loop only when no GPU is usable."""
    _check(lib().modgpu_cycle_auto_host(_host_ptr(buf), buf.size, as_int32(key), stream_off, device))
    return buf


def path_stats(reset=False):
    st = PathStats()
    _check(lib().modgpu_path_stats(ctypes.byref(st), 1 if reset else 0))
    return st.as_dict()


def gpu_required():
    return bool(lib().modgpu_gpu_required())


def last_launch():
    info = LaunchInfo()
    _check(lib().modgpu_last_launch(ctypes.byref(info)))
    return {"kernel": info.kernel.decode(), "variant": info.variant, "grid": info.grid, "block": info.block,
            "chunk_bytes": info.chunk_bytes, "bytes": info.bytes, "main_groups": info.main_groups,
            "source_hash": info.source_hash.decode() if info.source_hash else None}


SHAPES = {None: -1, "auto": -1, "small": 0, "large": 1, "queue": 2}


def debug_set_launch(shape=None, grid_cap=0):
    """Test hook: force the launch shape ("small" / "large" = streaming, static chunk map / "queue" = streaming,
    work queue / None = by size) and cap the grid."""
    _debug_lib().modgpu_debug_set_launch(SHAPES[shape], grid_cap or 0)


def debug_set_pinned_mode(mode=0):
    """Test hook: 0 default, 1 DMA pipeline, 2 kernel over PCIe, for pinned caller buffers."""
    _debug_lib().modgpu_debug_set_pinned_mode(mode)


def debug_set_staged_mode(mode=0):
    """Test hook: 0 default, 1 DMA, 2 kernel over PCIe on the pinned slot, for staged (pageable / file) chunks."""
    _debug_lib().modgpu_debug_set_staged_mode(mode)


HOST_TUNABLES = {"zerocopy_bytes": 0, "ring": 1, "split": 2, "chunk_min_bytes": 3, "ramp_bytes": 4, "lanes": 5, "ntcopy": 6, "file_sched": 7,
                 "feed": 8, "feed_chunk_bytes": 9, "feed_patience_ms": 10, "file_feed": 11}


def debug_set_host_tunable(name, value):
    """Testing flavour: one of HOST_TUNABLES at run time (not while a host-buffer call is in flight)."""
    _debug_lib().modgpu_debug_set_host_tunable(HOST_TUNABLES[name], int(value))


def debug_set_gpu_node(node=-2):
    """Testing flavour: the NUMA node the library believes its GPUs hang off (-2 = ask sysfs)."""
    _debug_lib().modgpu_debug_set_gpu_node(node)


def debug_set_pcie_grid(cap=0):
    """Measurement hook: workgroups of a launch across PCIe (0 = the product's rule)."""
    _debug_lib().modgpu_debug_set_pcie_grid(cap)


def debug_inject_failures(count):
    """Test hook: the next `count` host-buffer / file calls fail with MODGPU_ERR_HIP before touching anything."""
    _debug_lib().modgpu_debug_inject_failures(count)


STAGE_FILL, STAGE_LAUNCH, STAGE_SYNC, STAGE_DRAIN, STAGE_AFTER_DRAIN, STAGE_STALL = range(6)
INJECT_PIECE_LAST, INJECT_PIECE_MIDDLE = -1, -2


def debug_inject_failure_at(piece, stage):
    """Test hook: the HIP call of `stage` (STAGE_*) for piece `piece` of the next host-buffer / file call fails, once.  stage < 0 disarms."""
    _debug_lib().modgpu_debug_inject_failure_at(piece, stage)


def debug_forbid_worker_threads(forbid=True):
    """Testing flavour: staging sets start no worker thread from now on (as if thread creation failed)."""
    _debug_lib().modgpu_debug_forbid_worker_threads(1 if forbid else 0)


def debug_hold_slots(device, count):
    """Testing flavour: hold `count` pipeline slots of a device's staging set (0 releases); returns how many are held."""
    return _debug_lib().modgpu_debug_hold_slots(device, count)


def debug_injection_armed():
    return bool(_debug_lib().modgpu_debug_injection_armed())


def debug_set_helpers(mode=0):
    """Test hook: helper workgroups of the work-queue shape: 0 by the clock they measure, 1 always join, 2 none."""
    _debug_lib().modgpu_debug_set_helpers(mode)


def debug_set_queue_ring(lines=0):
    """Test hook: eager work-queue launches draw ticket pairs from the first `lines` ring lines only (0 = all 4096)."""
    _debug_lib().modgpu_debug_set_queue_ring(lines)


def debug_set_batch(mode=0):
    """Test hook: 0 several parts share a launch beyond 256 MiB in all or when small on average (shipped), 1 always, 2 never."""
    _debug_lib().modgpu_debug_set_batch(mode)


def queue_stats():
    """Work-queue bookkeeping since load (include/modgpu_testing.h: modgpu_queue_stats)."""
    out = (_u64 * 6)()
    lib().modgpu_queue_stats(out)
    return {"eager": int(out[0]), "busy_fallbacks": int(out[1]), "graph": int(out[2]), "graph_pool_empty": int(out[3]),
            "batch_launches": int(out[4]), "batch_parts": int(out[5])}


def testing_hooks():
    return bool(lib().modgpu_testing_hooks())


def min_gpu_bytes():
    return int(lib().modgpu_min_gpu_bytes())


def host_policy():
    """MODGPU_HOST_POLICY as latched: "offload" or "fastest"."""
    return lib().modgpu_host_policy().decode()


def host_policy_engine(n, pinned=False):
    """What `fastest` would decide for one call over n bytes: ("host" | "kernel", host_us, kernel_us)."""
    h, k = ctypes.c_double(0), ctypes.c_double(0)
    r = lib().modgpu_host_policy_engine(n, 1 if pinned else 0, ctypes.byref(h), ctypes.byref(k))
    return ("host" if r else "kernel"), h.value, k.value


def host_loop_isa():
    return lib().modgpu_host_loop_isa().decode()


def cycle_scalar_host_isa(buf, key, isa, stream_off=0):
    """The library's host loop with one named body ("generic" / "avx2" / "avx512")."""
    _check(lib().modgpu_cycle_scalar_host_isa(_host_ptr(buf), buf.size, as_int32(key), stream_off, isa.encode()))
    return buf


def device_numa_node(device):
    return lib().modgpu_device_numa_node(device)


def numa_probe(sysfs_root, bdf, max_cpus=4096):
    """(node, cpus) as the library reads them from a sysfs tree (tests hand it a fake one)."""
    node = _int(-1)
    cpus = (_int * max_cpus)()
    n = lib().modgpu_numa_probe(os.fsencode(sysfs_root), bdf.encode(), ctypes.byref(node), cpus, max_cpus)
    return node.value, list(cpus[:max(n, 0)])


def host_tunables():
    """The host-path tunables as latched (and clamped) at library load."""
    out = (_u64 * 4)()
    lib().modgpu_host_tunables(out)
    return {"pipes": int(out[0]), "chunk_bytes": int(out[1]), "zerocopy_max_bytes": int(out[2]), "ring": int(out[3])}


def host_chunking():
    out = (_u64 * 4)()
    lib().modgpu_host_chunking(out)
    return {"split": int(out[0]), "chunk_min_bytes": int(out[1]), "ramp_bytes": int(out[2]), "lanes": int(out[3])}


def host_trace(enable=True):
    """Start (and clear) / stop the host-side timeline of the host-buffer and file routes."""
    lib().modgpu_host_trace(1 if enable else 0)


def host_trace_read():
    """The recorded events as dicts, in recording order: t_ns, kind (HOST_TRACE_KINDS), pipe (-1: the call), chunk, bytes."""
    n = lib().modgpu_host_trace_read(None, 0)
    buf = (HostTraceEvent * max(n, 1))()
    n = min(lib().modgpu_host_trace_read(buf, n), n)
    return [{"t_ns": int(e.t_ns), "kind": HOST_TRACE_KINDS[e.kind], "pipe": e.pipe, "chunk": int(e.chunk), "bytes": int(e.bytes), "tid": e.tid} for e in buf[:n]]


def host_pool_stats():
    out = (_u64 * 6)()
    lib().modgpu_host_pool_stats(out)
    return {"workers_started": int(out[0]), "pipelines_run_by_workers": int(out[1]), "slot_waits": int(out[2]),
            "calls_overlapped": int(out[3]), "slots_per_device": int(out[4]), "calls_on_another_nodes_set": int(out[5])}


def prepare(device=-1):
    """modgpu_prepare: device preparation + the empty wake-up launch, for callers that bring their own device memory."""
    _check(lib().modgpu_prepare(device))


def kernel_source_hash():
    return lib().modgpu_kernel_source_hash().decode()


def feed_kernel_source_hash():
    """identity of the host-fed kernel's TU (cycle_feed_kernel.hip and what it includes)"""
    return lib().modgpu_feed_kernel_source_hash().decode()


class PinnedBuffer:
    """Host memory from modgpu_host_alloc, viewed as a numpy uint8 array (`.array`)."""

    def __init__(self, nbytes, near_device=None, parts=None, n_devices=0):
        """near_device: place the pages next to that GPU (modgpu_host_alloc_near); parts: list of part sizes laid end
        to end, part i next to GPU i mod n_devices (modgpu_host_alloc_parts; nbytes must be their sum)."""
        p = _vp()
        self._lib = lib()
        if parts is not None:
            assert sum(parts) == nbytes
            sizes = (_u64 * len(parts))(*parts)
            _check(self._lib.modgpu_host_alloc_parts(ctypes.byref(p), sizes, len(parts), n_devices))
        elif near_device is not None:
            _check(self._lib.modgpu_host_alloc_near(ctypes.byref(p), nbytes, near_device))
        else:
            _check(self._lib.modgpu_host_alloc(ctypes.byref(p), nbytes))
        self.ptr, self.nbytes = p.value, nbytes
        self.array = np.ctypeslib.as_array(ctypes.cast(self.ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(max(nbytes, 1),))[:nbytes]

    @property
    def pinned(self):
        return bool(self._lib.modgpu_host_is_pinned(_vp(self.ptr), self.nbytes))

    def free(self):
        if self.ptr:
            self.array = None
            _check(self._lib.modgpu_host_free(_vp(self.ptr)))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def host_register(buf):
    """Page-lock a caller-owned ndarray in place (modgpu_host_register); pair with host_unregister(buf)."""
    _check(lib().modgpu_host_register(_host_ptr(buf), buf.size))


def host_unregister(buf):
    _check(lib().modgpu_host_unregister(_host_ptr(buf)))


def hdr_decrypt_host(hdr, device=-1):
    _check(lib().modgpu_hdr_decrypt_host(_host_ptr(hdr), hdr.size, device))
    return hdr


def hdr_encrypt_host(hdr, ps4=True, device=-1):
    _check(lib().modgpu_hdr_encrypt_host(_host_ptr(hdr), hdr.size, 1 if ps4 else 0, device))
    return hdr


def cycle_parts_host(parts, key, n_devices=0):
    n = len(parts)
    ptrs = (_vp * n)(*[_host_ptr(p).value for p in parts])
    sizes = (_u64 * n)(*[p.size for p in parts])
    _check(lib().modgpu_cycle_parts_host(ptrs, sizes, n, as_int32(key), n_devices))
    return parts


def cycle_parts_device(buffers, key):
    """DeviceBuffers, each on its own device: every one cycled as its own stream from offset 0, all GPUs at once."""
    n = len(buffers)
    ptrs = (_vp * n)(*[b.ptr for b in buffers])
    sizes = (_u64 * n)(*[b.nbytes for b in buffers])
    devs = (_int * n)(*[b.device for b in buffers])
    _check(lib().modgpu_cycle_parts_device(ptrs, sizes, devs, n, as_int32(key)))


def cycle_host_split(buf, key, stream_off=0, n_devices=0):
    """One host buffer over several GPUs: contiguous spans, span d on GPU d with its own stream offset (no exchange step)."""
    _check(lib().modgpu_cycle_host_split(_host_ptr(buf), buf.size, as_int32(key), stream_off, n_devices))
    return buf


def cycle_batch_device(ptrs, sizes, key, stream_offs=None, device=-1, stream=None):
    """Raw device addresses of ONE device, each its own Cycle call (from stream_offs[i] or 0); asynchronous on `stream`.
    Runs of up to 16 parts share one kernel launch when together they are beyond 256 MiB or small on average."""
    n = len(ptrs)
    p = (_vp * n)(*ptrs)
    z = (_u64 * n)(*sizes)
    o = (_u64 * n)(*stream_offs) if stream_offs is not None else None
    _check(lib().modgpu_cycle_batch_device(p, z, o, n, as_int32(key), device, _vp(stream or 0)))


def cycle_file(src_path, dst_path, key, stream_off=0, device=-1):
    """Stream a whole part file through the GPU (dst may equal src: in place)."""
    _check(lib().modgpu_cycle_file(os.fsencode(src_path), os.fsencode(dst_path), as_int32(key), stream_off, device))


def cycle_file_to_host(path, n, key, file_off=0, stream_off=0, device=-1, out=None):
    """n bytes of a part file through the GPU into host memory (`out`: a caller array, e.g. a view of a PinnedBuffer)."""
    if out is None:
        out = np.empty(n, dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size == n and out.flags["C_CONTIGUOUS"]
    _check(lib().modgpu_cycle_file_to_host(os.fsencode(path), file_off, _vp(out.ctypes.data), n, as_int32(key),
                                           stream_off, device))
    return out


def cycle_host_to_file(buf, path, key, stream_off=0, device=-1):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    _check(lib().modgpu_cycle_host_to_file(_vp(buf.ctypes.data), buf.size, os.fsencode(path), as_int32(key),
                                           stream_off, device))


def cycle_device(dev_ptr, n, key, stream_off=0, device=-1, stream=None):
    """Asynchronous in-place cycle of n device-resident bytes at raw address dev_ptr."""
    _check(lib().modgpu_cycle_device(_vp(dev_ptr), n, as_int32(key), stream_off, device, _vp(stream or 0)))


def time_cycle_device(dev_ptr, n, key, stream_off=0, device=-1, stream=None, iters=2):
    """Mean ms per launch over `iters` launches, HIP events on the launch stream."""
    ms = ctypes.c_float(0)
    _check(lib().modgpu_time_cycle_device(_vp(dev_ptr), n, as_int32(key), stream_off, device, _vp(stream or 0),
                                          iters, ctypes.byref(ms)))
    return ms.value


def state_at(key, i):
    return lib().modgpu_state_at(as_int32(key), i)


def jump_table(which):
    out = (ctypes.c_uint32 * 256)()
    n = lib().modgpu_jump_table(which, out, 256)
    return list(out[:n])


class DeviceBuffer:
    """hipMalloc'd bytes owned through modgpu_alloc / modgpu_free."""

    def __init__(self, nbytes, device=-1):
        self.nbytes, self.device = nbytes, device
        p = _vp()
        self._lib = lib()
        _check(self._lib.modgpu_alloc(ctypes.byref(p), nbytes, device))
        self.ptr = p.value

    def upload(self, host, offset=0):
        host = np.ascontiguousarray(host, dtype=np.uint8)
        assert offset + host.size <= self.nbytes
        _check(self._lib.modgpu_h2d(_vp(self.ptr + offset), _vp(host.ctypes.data), host.size, self.device))

    def download(self, n=None, offset=0):
        n = self.nbytes - offset if n is None else n
        out = np.empty(n, dtype=np.uint8)
        _check(self._lib.modgpu_d2h(_vp(out.ctypes.data), _vp(self.ptr + offset), n, self.device))
        return out

    def cycle(self, key, n=None, offset=0, stream_off=0, stream=None):
        n = self.nbytes - offset if n is None else n
        assert offset + n <= self.nbytes
        cycle_device(self.ptr + offset, n, key, stream_off, self.device, stream)

    def sync(self, stream=None):
        _check(self._lib.modgpu_sync(self.device, _vp(stream or 0)))

    def free(self):
        if self.ptr:
            _check(self._lib.modgpu_free(_vp(self.ptr), self.device))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
