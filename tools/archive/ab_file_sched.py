#!/usr/bin/env python3
"""tools/archive/ab_file_sched.py -- file -> page-locked memory (LoadArkData's part cipher) and page-locked memory -> file (SaveArk's) with the file
routes' own schedule against the memory routes' schedule (testing flavour, interleaved, best of the repetitions; tmpfs)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import modulate_amd as M  # noqa: E402

M.use_testing_flavour()
d = "/dev/shm"
rng = np.random.default_rng(3)
res = {}
for mib in (64, 392, 4096):
    n = mib << 20
    path = os.path.join(d, f"ab_file_sched_{mib}.part")
    np.resize(rng.integers(0, 256, size=min(n, 1 << 26), dtype=np.uint8), n).tofile(path)
    pb = M.PinnedBuffer(n)
    for rep in range(4):
        for sched in (0, 1):
            M.debug_set_host_tunable("file_sched", sched)
            t0 = time.perf_counter()
            M.cycle_file_to_host(path, n, M.KEY_PS4, out=pb.array)
            t = time.perf_counter() - t0
            k = ("file -> page-locked", mib, sched)
            res[k] = min(res.get(k, 1e9), t)
    out = path + ".out"
    for rep in range(3):
        for sched in (0, 1):
            M.debug_set_host_tunable("file_sched", sched)
            t0 = time.perf_counter()
            M.cycle_host_to_file(pb.array, out, M.KEY_PS4)
            t = time.perf_counter() - t0
            k = ("page-locked -> file", mib, sched)
            res[k] = min(res.get(k, 1e9), t)
    os.unlink(out)
    os.unlink(path)
    pb.free()
print("GB/s of payload, best call; sched 0 = the file routes' own schedule, 1 = file -> memory takes the memory routes' (finer cut, ramp, lanes)")
for (route, mib, sched), t in sorted(res.items()):
    print("  %-20s %5d MiB  sched %d  %6.2f GB/s" % (route, mib, sched, (mib << 20) / t / 1e9))
