#!/usr/bin/env python3
"""Sweep the host-buffer path's tunables (each setting in a fresh process: they are read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import modulate_amd as M
M.use_testing_flavour()  # the staging knobs set through the environment below exist in libmodgpu_testing.so only (round 5)
for n in (64 << 20, 1 << 30, 1 << 32):
    buf = np.empty(n, np.uint8); buf[:] = 7
    M.cycle_host(buf, M.KEY_PS4)            # warm: staging allocation, page faults
    reps = 5 if n < (1 << 30) else 3
    t0 = time.perf_counter()
    for _ in range(reps): M.cycle_host(buf, M.KEY_PS4)
    dt = (time.perf_counter() - t0) / reps
    print(f"   n={n>>20:5d} MiB  {dt*1e3:8.2f} ms  {n/dt/1e9:6.1f} GB/s", flush=True)
''' % ROOT
for pipes in (1, 2, 4, 6, 8, 12, 16):
    for chunk in (4, 8, 16, 32):
        env = dict(os.environ, MODGPU_HOST_PIPES=str(pipes), MODGPU_HOST_CHUNK_MB=str(chunk))
        print(f"pipes={pipes} chunk={chunk} MiB", flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
