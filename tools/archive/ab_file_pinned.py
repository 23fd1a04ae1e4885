#!/usr/bin/env python3
"""tools/archive/ab_file_pinned.py (round 6; the library it measured had all three ways behind its testing knob -- the two that lost have been
removed, so this no longer runs against the tree: kept as the record of how profiles/r06_file_routes.txt's last table was taken)

tools/ab_file_pinned.py [reps=5] [MiB,...] -- FILE -> PAGE-LOCKED memory (CArk::LoadArkData's part cipher): the three ways, interleaved:
launch per chunk in the destination (round 5) / ONE host-fed kernel in the destination / ONE host-fed kernel through the staging slots with
a copy out (what a pageable destination takes).  Testing flavour; tmpfs; results compared whole against the library's host loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import modulate_amd as M  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 392, 1024, 2040]
socket = sys.argv[3] if len(sys.argv) > 3 else "any"  # near | far | any: this process (and so the file's pages on tmpfs) on the GPU's NUMA node or another


def bind(which):
    node = M.device_numa_node(0)
    if which == "any" or node < 0:
        return "unplaced"
    for k in range(64):
        if (k == node) != (which == "near"):
            continue
        try:
            text = open(f"/sys/devices/system/node/node{k}/cpulist").read().strip()
        except OSError:
            continue
        cpus = set()
        for part in text.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return f"on NUMA node {k} ({which}; the GPU hangs off node {node})"
    return "unplaced (no such node)"


M.use_testing_flavour()
print("this process:", bind(socket))
M.debug_set_host_tunable("feed_inplace_below_mb", 1 << 20)
rng = np.random.default_rng(3)
rows = [("launch per chunk, in place", 0), ("host-fed, in place", 1), ("host-fed, through the slots", 2)]
print(f"file (tmpfs) -> page-locked memory, {reps} calls per row, interleaved; GB/s of payload: best / median")
for mib in sizes:
    n = mib << 20
    path = f"/dev/shm/ab_file_pinned_{os.getpid()}_{mib}.part"
    pt = np.resize(rng.integers(0, 256, size=min(n, 1 << 26), dtype=np.uint8), n)
    pt.tofile(path)
    want = pt.copy()
    M.cycle_scalar_host(want, M.KEY_PS4)
    del pt
    pb = M.PinnedBuffer(n + 64)
    dst = pb.array[4:4 + n]
    times = {r[0]: [] for r in rows}
    for rep in range(reps + 1):
        for name, mode in rows:
            M.debug_set_host_tunable("file_feed", mode)
            dst[:4096] = 0
            t0 = time.perf_counter()
            M.cycle_file_to_host(path, n, M.KEY_PS4, out=dst)
            t = time.perf_counter() - t0
            if rep >= 1:
                times[name].append(t)
            else:
                assert np.array_equal(dst, want), (mib, name)
    print(f"  {mib:5d} MiB  " + "   ".join(f"{name}: {n / min(ts) / 1e9:5.2f} / {n / sorted(ts)[len(ts) // 2] / 1e9:5.2f}" for name, ts in times.items()), flush=True)
    os.unlink(path)
    pb.free()
M.debug_set_host_tunable("file_feed", 1)
