#!/usr/bin/env python3
"""A caller that brings its OWN device memory: hipMalloc + hipMemcpy of its own (no modgpu_alloc, no modgpu_h2d), then the
process's first launch over the part.  With `--prepare` it calls modgpu_prepare(device) when its upload starts -- the housekeeping
modgpu_alloc / modgpu_h2d do for callers who upload through them (include/modgpu.h) -- without it the first launch pays for the
code object, the ticket ring and the sleeping shader engines.  Run in a FRESH process each time (that is the point); prints one JSON
line: the first launch (one pair of HIP events on the launch stream), the next one, and the steady rate of the same size.

    python3 tools/first_launch_own_upload.py [--prepare | --library-upload] [--bytes 411000000] [--idle-ms 1500]

--library-upload: the other kind of caller, for comparison in the same harness: modgpu_alloc + modgpu_h2d instead of hipMalloc + hipMemcpy.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MODGPU_REQUIRE_GPU"] = "1"
import numpy as np  # noqa: E402

import modulate_amd as M  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prepare", action="store_true")
    ap.add_argument("--library-upload", action="store_true")
    ap.add_argument("--bytes", type=int, default=411 * 1000 * 1000)
    ap.add_argument("--idle-ms", type=int, default=1500, help="pause before the upload: the shader engines are asleep by then")
    a = ap.parse_args()
    assert M.device_count() >= 1
    hip = ctypes.CDLL("libamdhip64.so")  # the runtime libmodgpu.so already brought in
    n = a.bytes
    dev = ctypes.c_void_p()
    assert hip.hipSetDevice(0) == 0
    tile = np.random.default_rng(1).integers(0, 256, size=min(n, 64 << 20), dtype=np.uint8)
    if a.library_upload:
        lib_buf = M.DeviceBuffer(n, device=0)
        dev.value = lib_buf.ptr
    else:
        assert hip.hipMalloc(ctypes.byref(dev), ctypes.c_size_t(n)) == 0
    time.sleep(a.idle_ms / 1000.0)
    if a.prepare:
        M.prepare(0)  # when the upload STARTS: the engines wake while the DMA engine works
    for off in range(0, n, tile.size):
        ln = min(tile.size, n - off)
        if a.library_upload:
            lib_buf.upload(tile[:ln], offset=off)
        else:
            assert hip.hipMemcpy(ctypes.c_void_p(dev.value + off), ctypes.c_void_p(tile.ctypes.data), ctypes.c_size_t(ln), 1) == 0
    first = M.time_cycle_device(dev.value, n, M.KEY_PS4, 0, 0, None, iters=1)
    nxt = M.time_cycle_device(dev.value, n, M.KEY_PS4, 0, 0, None, iters=1)
    M.time_cycle_device(dev.value, n, M.KEY_PS4, 0, 0, None, iters=4)
    steady = M.time_cycle_device(dev.value, n, M.KEY_PS4, 0, 0, None, iters=20)
    got = np.empty(min(n, 1 << 20), np.uint8)
    assert hip.hipMemcpy(ctypes.c_void_p(got.ctypes.data), dev, ctypes.c_size_t(got.size), 2) == 0
    ok = bool(np.array_equal(got, tile[:got.size]))  # an even number of passes: the plaintext again

    def rate(ms):
        return {"ms": round(ms, 4), "frac_of_8TBs": round(2.0 * n / (ms * 1e-3) / 1e9 / 8000.0, 4)}
    print(json.dumps({"prepare": a.prepare, "library_upload": a.library_upload, "bytes": n, "first_launch": rate(first), "next_launch": rate(nxt), "steady": rate(steady),
                      "first_over_steady": round(first / steady, 4), "involution_ok": ok, "kernel": M.last_launch()["kernel"]}))
    if a.library_upload:
        lib_buf.free()
    else:
        hip.hipFree(dev)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
