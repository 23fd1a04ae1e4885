#!/usr/bin/env python3
"""Where a single buffer should change from the small shape to the work-queue shape (kLargeMin in modgpu_capi.cpp), measured
on the shipped kernels through the library's testing flavour: each size with the shape forced, one launch timed by HIP events,
(a) cold -- 1 GiB of other data cycled in between, so the buffer is out of the Infinity Cache but the chip is busy and its
clock settled -- and (b) warm, the same buffer again right away.  Median of `rounds`.

    python tools/archive/handover.py [rounds=9]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MODGPU_REQUIRE_GPU"] = "1"
import numpy as np  # noqa: E402
import modulate_amd as M  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 9
M.use_testing_flavour()
other = M.DeviceBuffer(1 << 30)
sizes = [16 << 20, 32 << 20, 64 << 20, 96 << 20, 128 << 20, 160 << 20, 192 << 20, 224 << 20, 256 << 20, 320 << 20, 384 << 20, 512 << 20]
buf = M.DeviceBuffer(max(sizes) + 64)
print("%10s | %21s | %21s |  (TB/s read+write; cold / warm)" % ("bytes", "small shape", "work-queue shape"))
for n in sizes:
    row = {}
    for shape in ("small", "queue"):
        cold, warm = [], []
        for _ in range(rounds):
            M.debug_set_launch(None)
            other.cycle(M.KEY_PS3)
            other.cycle(M.KEY_PS3)
            M.debug_set_launch(shape)
            cold.append(M.time_cycle_device(buf.ptr + 4, n, M.KEY_PS4, 0, 0, None, iters=1))
            assert M.last_launch()["variant"] == {"small": 0, "queue": 2}[shape]
            warm.append(M.time_cycle_device(buf.ptr + 4, n, M.KEY_PS4, 0, 0, None, iters=1))
        row[shape] = (2.0 * n / (sorted(cold)[rounds // 2] * 1e-3) / 1e12, 2.0 * n / (sorted(warm)[rounds // 2] * 1e-3) / 1e12)
    best = "queue" if row["queue"][0] > row["small"][0] else "small"
    print("%10d | %9.3f / %9.3f | %9.3f / %9.3f |  cold: %s" % (n, *row["small"], *row["queue"], best))
