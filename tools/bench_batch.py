#!/usr/bin/env python3
"""Several resident parts per pass: one launch per part against modgpu_cycle_batch_device (A/B inside one process, on the
testing flavour of the library, which can switch batching off).  Wall clock around `steps` passes, one wait at the end.

    python tools/bench_batch.py [--steps 40]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MODGPU_REQUIRE_GPU"] = "1"
import numpy as np  # noqa: E402
import modulate_amd as M  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    M.use_testing_flavour()
    shapes = [(8, 411_000_000), (2, 411_000_000), (4, 411_000_000), (16, 411_000_000), (8, 100_000_000), (16, 50_000_000),
              (4, 1 << 30), (2, 1 << 32), (8, 33_554_432 + 4),
              # under the shipped threshold (256 MiB in all): forced, to see where the threshold belongs
              (2, 64 << 20), (4, 32 << 20), (8, 16 << 20), (16, 8 << 20), (4, 16 << 20), (8, 4 << 20), (16, 1 << 20), (4, 1 << 20), (8, 65536)]
    tile = np.random.default_rng(1).integers(0, 256, size=1 << 24, dtype=np.uint8)
    print("%5s %12s | %21s | %21s | %6s" % ("parts", "bytes each", "one launch per part", "batched", "gain"))
    print("%5s %12s | %9s %11s | %9s %11s |" % ("", "", "ms/pass", "TB/s r+w", "ms/pass", "TB/s r+w"))
    for n_parts, n in shapes:
        bufs = [M.DeviceBuffer(n + 16) for _ in range(n_parts)]
        for b in bufs:
            for off in range(0, n, tile.size):
                b.upload(tile[:min(tile.size, n - off)], offset=off)
        ptrs, sizes = [b.ptr + 4 for b in bufs], [n] * n_parts  # buf+4, like the reference's callers
        row = {}
        for mode in (2, 1, 2, 1):
            M.debug_set_batch(mode)
            for _ in range(5):
                M.cycle_batch_device(ptrs, sizes, M.KEY_PS4, device=0)
            bufs[0].sync()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                M.cycle_batch_device(ptrs, sizes, M.KEY_PS4, device=0)
            bufs[0].sync()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            row[mode] = min(row.get(mode, 1e9), ms)
        M.debug_set_batch(0)
        M.cycle_batch_device(ptrs, sizes, M.KEY_PS4, device=0)
        M.cycle_batch_device(ptrs, sizes, M.KEY_PS4, device=0)
        bufs[0].sync()
        kind = M.last_launch()["variant"]
        tb = lambda ms: 2.0 * n * n_parts / (ms * 1e-3) / 1e12
        print("%5d %12d | %9.4f %11.3f | %9.4f %11.3f | %+5.1f%%%s" % (n_parts, n, row[2], tb(row[2]), row[1], tb(row[1]),
              100 * (row[2] / row[1] - 1), "" if kind == 3 else "   (shipped rule: not batched)"))
        # an even number of passes in all: the parts hold their input again
        w = min(n, 1 << 16)
        assert np.array_equal(bufs[-1].download(w, offset=n - w), np.resize(np.roll(tile, -((n - w) % tile.size)), w))
        for b in bufs:
            b.free()
    print(json.dumps(M.queue_stats()))


if __name__ == "__main__":
    main()
