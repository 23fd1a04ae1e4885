#!/usr/bin/env python3
"""tools/ab_feed.py [reps=10] -- the pageable (staged) host route with ONE host-fed kernel per call (cycle_feed_kernel.h) against a kernel launch per
chunk (the schedule until round 5), through the library itself: testing flavour, settings interleaved call by call, pageable buffers, results
checked against the library's host loop once per size.  GB/s of payload: best and median call.  -> profiles/r05_pcie_feed.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import modulate_amd as M  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
M.use_testing_flavour()
rng = np.random.default_rng(5)
rows = [("launch per chunk", 0, 256), ("host-fed, 256 KiB chunks", 1, 256), ("host-fed, 128 KiB chunks", 1, 128), ("host-fed, 512 KiB chunks", 1, 512), ("host-fed, 1 MiB chunks", 1, 1024)]
print(f"pageable buffer, modgpu_cycle_host, {reps} calls per row and size, interleaved; GB/s of payload: best / median")
for mib in (4, 8, 16, 32, 64, 256, 1024, 4096):
    n = mib << 20
    pt = np.resize(rng.integers(0, 256, size=min(n, 1 << 26), dtype=np.uint8), n)
    buf = pt.copy()
    times = {r[0]: [] for r in rows}
    for rep in range(reps + 2):
        for name, feed, kb in rows:
            M.debug_set_host_tunable("feed", feed)
            M.debug_set_host_tunable("feed_chunk_bytes", kb << 10)
            t0 = time.perf_counter()
            M.cycle_host(buf, M.KEY_PS4)
            t = time.perf_counter() - t0
            if rep >= 2:
                times[name].append(t)
    # an even number of passes per row and an odd number of rows x passes in all?  check against the host loop instead of counting
    passes = (reps + 2) * len(rows)
    want = pt.copy()
    if passes % 2:
        M.cycle_scalar_host(want, M.KEY_PS4)
    assert np.array_equal(buf, want), f"{mib} MiB: result differs from the host loop"
    print(f"  {mib:5d} MiB  " + "   ".join(f"{name}: {n / min(ts) / 1e9:5.2f} / {n / sorted(ts)[len(ts) // 2] / 1e9:5.2f}" for name, ts in times.items()), flush=True)
M.debug_set_host_tunable("feed", 1)
M.debug_set_host_tunable("feed_chunk_bytes", 256 << 10)
