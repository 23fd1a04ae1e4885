#!/usr/bin/env python3
"""Per-call latency of modgpu_cycle_host on header-sized buffers, zero-copy path on vs off
(MODGPU_HOST_ZEROCOPY_KB is read once, so each setting runs in its own process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import modulate_amd as M
M.use_testing_flavour()  # the staging knobs set through the environment below exist in libmodgpu_testing.so only (round 5)
for n in (64, 4096, 65536, 262144, 524288, 1 << 20, 2 << 20):
    buf = np.random.default_rng(1).integers(0, 256, size=n, dtype=np.uint8)
    for _ in range(20): M.cycle_host(buf, M.KEY_PS4)
    t0 = time.perf_counter()
    for _ in range(400): M.cycle_host(buf, M.KEY_PS4)
    print(f"   n={n:8d}  {(time.perf_counter()-t0)/400*1e6:8.1f} us/call", flush=True)
''' % ROOT
for kb in (0, 1024):
    print(f"MODGPU_HOST_ZEROCOPY_KB={kb}", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MODGPU_HOST_ZEROCOPY_KB=str(kb)))
