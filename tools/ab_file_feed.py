#!/usr/bin/env python3
"""tools/ab_file_feed.py [reps=5] -- FILE -> memory (LoadArkData's part cipher, Modulate/CArk.cpp:741-755) with ONE host-fed kernel per call
(round 6) against round 5's launch per chunk, through the library itself: testing flavour, the two settings interleaved call by call, the file
on tmpfs.  Into pageable memory (pread replaces the copy into the slot) and into page-locked memory (in place: the chunks are read to where
they belong).  Results compared whole against the library's host loop once per size and destination.  GB/s of payload: best / median call.
The last rows: the feed chunk size for file sources.  -> profiles/r06_file_routes.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import modulate_amd as M  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 392, 4096]
M.use_testing_flavour()
d = "/dev/shm"
rng = np.random.default_rng(3)
print(f"file (tmpfs) -> memory, modgpu_cycle_file_to_host, {reps} calls per row, settings interleaved; GB/s of payload: best / median")
for mib in sizes:
    n = mib << 20
    path = os.path.join(d, f"ab_file_feed_{os.getpid()}_{mib}.part")
    pt = np.resize(rng.integers(0, 256, size=min(n, 1 << 26), dtype=np.uint8), n)
    pt.tofile(path)
    want = pt.copy()
    M.cycle_scalar_host(want, M.KEY_PS4)
    del pt
    pb = M.PinnedBuffer(n + 64)
    pageable = np.zeros(n + 64, np.uint8)
    rows = [("launch per chunk", 0, 256), ("host-fed 256 KiB", 1, 256), ("host-fed 512 KiB", 1, 512), ("host-fed 1 MiB", 1, 1024), ("host-fed 2 MiB", 1, 2048)]
    for dst_name, dst in (("pageable   ", pageable[4:4 + n]), ("page-locked", pb.array[4:4 + n])):
        times = {r[0]: [] for r in rows}
        for rep in range(reps + 1):
            for name, feed, kb in rows:
                M.debug_set_host_tunable("file_feed", feed)
                M.debug_set_host_tunable("feed_chunk_bytes", kb << 10)
                dst[:4096] = 0
                t0 = time.perf_counter()
                M.cycle_file_to_host(path, n, M.KEY_PS4, out=dst)
                t = time.perf_counter() - t0
                if rep >= 1:
                    times[name].append(t)
                if rep == 0:
                    assert np.array_equal(dst, want), f"{mib} MiB -> {dst_name}: {name}: result differs from the host loop"
        print(f"  {mib:5d} MiB -> {dst_name}  " + "   ".join(f"{name}: {n / min(ts) / 1e9:5.2f} / {n / sorted(ts)[len(ts) // 2] / 1e9:5.2f}" for name, ts in times.items()), flush=True)
    os.unlink(path)
    pb.free()
M.debug_set_host_tunable("file_feed", 1)
M.debug_set_host_tunable("feed_chunk_bytes", 256 << 10)
