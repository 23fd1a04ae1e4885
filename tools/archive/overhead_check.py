"""tools/archive/overhead_check.py -- wall-clock minus HIP-event time of modgpu_time_cycle_device (the timed region of
bench.py): ~15-40 us per call, i.e. bench.py's ms_per_step is kernel time, not host overhead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import modulate_amd as M
d = M.DeviceBuffer(1 << 32)
for n, iters in ((4096, 2), (4096, 40), (1 << 32, 2), (1 << 32, 10), (1 << 32, 40)):
    M.time_cycle_device(d.ptr, n, M.KEY_PS4, iters=2)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        ms = M.time_cycle_device(d.ptr, n, M.KEY_PS4, iters=iters)
        d.sync()
        dt = (time.perf_counter() - t0) * 1e3
        best = min(best, dt - ms * iters)
    print(f"n={n:>11} iters={iters:3d}: events {ms*iters:9.3f} ms, wall-minus-events overhead (best of 5) {best*1e3:8.1f} us")
