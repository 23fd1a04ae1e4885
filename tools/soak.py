#!/usr/bin/env python3
"""tools/soak.py [seconds=120] [seed=1] -- randomized long-run check of the streaming kernels (not part of the test-suite).

Random sizes, misalignments, stream offsets and keys on the work-queue shape with forced small grids, helper workgroups forced to
join / absent / by the clock, a ticket ring of 1-3 lines and two streams in flight; every buffer is compared whole against the
library's own host loop (itself pinned to the oracle and the golden vectors by tests/test_capi_cpu.py), every 16th also against
the oracle.  Prints a summary line; any mismatch raises."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.pop("MODGPU_REQUIRE_GPU", None)  # the host loop is the checker here (a different engine from the one under test)
import modulate_amd as M  # noqa: E402
import hip_rt  # noqa: E402
from oracle import oracle as O  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
M.use_testing_flavour()
assert M.device_count() >= 1
cap = (96 << 20) + 8192
bufs = [M.DeviceBuffer(cap), M.DeviceBuffer(cap)]
streams = [hip_rt.Stream(), hip_rt.Stream()]
t_end = time.time() + seconds
n_cases = n_bytes = 0
while time.time() < t_end:
    shape = ["queue", "queue", "large", None][int(rng.integers(0, 4))]
    grid = int(rng.choice([1, 2, 3, 5, 8, 13, 24, 64, 200, 256]))
    M.debug_set_launch(shape, grid if shape else 0)
    M.debug_set_helpers(int(rng.integers(0, 3)))
    M.debug_set_queue_ring(int(rng.integers(0, 4)))
    jobs = []
    for b, st in zip(bufs, streams):
        n = int(rng.integers(1, 96 << 20)) if rng.random() < 0.7 else int(rng.choice([65536 * 3, 65536 * 7 + 5, 131072 * 9 - 16, 1 << 20]))
        base = int(rng.integers(0, 4096))
        key = int(rng.choice([0x90CFC0AB, 0xC64EED30, int(rng.integers(1, 1 << 32))]))
        so = int(rng.choice([0, int(rng.integers(0, 1 << 62)), O.PERIOD - n // 3]))
        passes = int(rng.choice([1, 3]))
        pt = rng.integers(0, 256, size=n + 64, dtype=np.uint8)
        b.upload(pt, offset=max(0, base - 32))
        lo = max(0, base - 32)
        for _ in range(passes):
            b.cycle(key, n=n, offset=base, stream_off=so, stream=st.handle)
        jobs.append((b, st, n, base, lo, key, so, pt))
    for b, st, n, base, lo, key, so, pt in jobs:
        st.sync()
        want = pt.copy()
        view = want[base - lo:base - lo + n]
        M.cycle_scalar_host(view, key, so)
        got = b.download(n + 64, offset=lo)
        if not np.array_equal(got, want):
            bad = np.flatnonzero(got != want)
            raise SystemExit(f"MISMATCH shape={shape} grid={grid} n={n} base={base} key={key:#x} so={so}: {bad.size} bytes differ, first at {bad[0]}")
        if n_cases % 16 == 0 and n <= (8 << 20):
            w2 = pt.copy()
            O.cycle_at(w2[base - lo:base - lo + n], key, so)
            assert np.array_equal(want, w2), "host loop vs oracle"
        n_cases += 1
        n_bytes += n
q = M.queue_stats()
print(f"SOAK_OK {n_cases} buffers, {n_bytes / 1e9:.1f} GB cycled and compared in {seconds:.0f} s; work-queue bookkeeping {q}")
