#!/usr/bin/env python3
"""Device-resident throughput of modgpu_cycle_device as a function of buffer size (which launch
shape the host layer picks, and where the hand-over between them sits)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import modulate_amd as M  # noqa: E402
M.use_testing_flavour()  # modgpu_debug_set_launch exists only in libmodgpu_testing.so

cap = 1 << 30
if len(sys.argv) > 1:  # force a launch shape: small | large | queue
    M.debug_set_launch(sys.argv[1], 0)
d = M.DeviceBuffer(cap + 64)
d.upload(np.zeros(1 << 20, np.uint8))
print(f"{'bytes':>12} {'us/launch':>10} {'GB/s r+w':>10}   (aligned base | base+4)")
for sh in range(12, 31):
    for n in (1 << sh, 3 << (sh - 1)):
        if n > cap:
            continue
        row = []
        for off in (0, 4):
            iters = 200 if n <= (16 << 20) else 20
            M.time_cycle_device(d.ptr + off, n, M.KEY_PS4, iters=4)
            ms = M.time_cycle_device(d.ptr + off, n, M.KEY_PS4, iters=iters)
            row.append((ms * 1e3, 2 * n / (ms * 1e-3) / 1e9))
        print(f"{n:12d} {row[0][0]:10.1f} {row[0][1]:10.1f}   | {row[1][0]:10.1f} {row[1][1]:10.1f}")
