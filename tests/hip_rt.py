"""The few HIP runtime calls the GPU tests need that the product ABI does not (and should not) offer:
streams and stream capture into a hipGraph.  ctypes over the HIP runtime that libmodgpu.so already
brought into the process -- test plumbing, nothing is computed here."""
import ctypes
import ctypes.util
import os

_vp = ctypes.c_void_p
_hip = None


def hip():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", ctypes.util.find_library("amdhip64") or "", "/opt/rocm/lib/libamdhip64.so"):
            if not name:
                continue
            try:
                _hip = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("HIP runtime not found")
        _hip.hipGetErrorString.restype = ctypes.c_char_p
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc} ({hip().hipGetErrorString(rc).decode()})")


class Stream:
    NON_BLOCKING = 1

    def __init__(self, flags=NON_BLOCKING):
        h = _vp()
        _ok(hip().hipStreamCreateWithFlags(ctypes.byref(h), ctypes.c_uint(flags)), "hipStreamCreateWithFlags")
        self.handle = h.value

    def sync(self):
        _ok(hip().hipStreamSynchronize(_vp(self.handle)), "hipStreamSynchronize")

    def destroy(self):
        if self.handle:
            _ok(hip().hipStreamDestroy(_vp(self.handle)), "hipStreamDestroy")
            self.handle = None


class Graph:
    """`with Graph.capture(stream) as g:` records what is launched on `stream` in the block; g.launch(stream) replays it."""
    MODE_THREAD_LOCAL = 1

    def __init__(self):
        self.graph = _vp()
        self.exe = _vp()

    class _Cap:
        def __init__(self, g, stream):
            self.g, self.stream = g, stream

        def __enter__(self):
            _ok(hip().hipStreamBeginCapture(_vp(self.stream.handle), ctypes.c_int(Graph.MODE_THREAD_LOCAL)), "hipStreamBeginCapture")
            return self.g

        def __exit__(self, *exc):
            _ok(hip().hipStreamEndCapture(_vp(self.stream.handle), ctypes.byref(self.g.graph)), "hipStreamEndCapture")
            if exc[0] is None:
                _ok(hip().hipGraphInstantiate(ctypes.byref(self.g.exe), self.g.graph, None, None, ctypes.c_size_t(0)), "hipGraphInstantiate")
            return False

    @classmethod
    def capture(cls, stream):
        return cls._Cap(cls(), stream)

    def launch(self, stream):
        _ok(hip().hipGraphLaunch(self.exe, _vp(stream.handle)), "hipGraphLaunch")

    def destroy(self):
        if self.exe:
            hip().hipGraphExecDestroy(self.exe)
        if self.graph:
            hip().hipGraphDestroy(self.graph)
        self.exe = self.graph = _vp()


def device_sync():
    _ok(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
