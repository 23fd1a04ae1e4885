#!/usr/bin/env python3
"""tools/soak.py [seconds=120] [seed=1] -- randomized long-run check of the streaming kernels (not part of the test-suite).

Random sizes, misalignments, stream offsets and keys on the work-queue shape with forced small grids, helper workgroups forced to
join / absent / by the clock, a ticket ring of 1-3 lines and two streams in flight, and a third of the time 2-20 parts of a
buffer through modgpu_cycle_batch_device (batching by the shipped rule / forced / off); every buffer is compared whole against the
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
n_cases = n_bytes = n_parts = 0
t_note = time.time() + 60
while time.time() < t_end:
    if time.time() > t_note:  # (a run that prints nothing for minutes looks hung to the box's watchdog)
        print(f"... {n_cases} buffers, {n_bytes / 1e9:.1f} GB so far", flush=True)
        t_note = time.time() + 60
    shape = ["queue", "queue", "large", None][int(rng.integers(0, 4))]
    grid = int(rng.choice([1, 2, 3, 5, 8, 13, 24, 64, 200, 256]))
    M.debug_set_launch(shape, grid if shape else 0)
    M.debug_set_helpers(int(rng.integers(0, 3)))
    M.debug_set_queue_ring(int(rng.integers(0, 4)))
    M.debug_set_batch(int(rng.choice([0, 1, 1, 2])))
    jobs = []
    for b, st in zip(bufs, streams):
        key = int(rng.choice([0x90CFC0AB, 0xC64EED30, int(rng.integers(1, 1 << 32))]))
        passes = int(rng.choice([1, 3]))
        if rng.random() < 0.35:  # several parts of this buffer through modgpu_cycle_batch_device (one launch for up to 16 of them)
            k = int(rng.integers(2, 21))
            room = (cap - 8192) // k
            segs = []
            for i in range(k):
                n = int(rng.choice([0, int(rng.integers(1, 64)), int(rng.integers(1, room - 64)), 65536 * int(rng.integers(1, 4)), int(rng.integers(1, 300_000))]))
                n = min(n, room - 64)
                segs.append((i * room + int(rng.integers(0, 48)), n, int(rng.choice([0, int(rng.integers(0, 1 << 62)), O.PERIOD - n // 3]))))
            lo, hi = 0, k * room
            pt = rng.integers(0, 256, size=hi - lo, dtype=np.uint8)
            b.upload(pt, offset=lo)
            for _ in range(passes):
                M.cycle_batch_device([b.ptr + o for o, _, _ in segs], [n for _, n, _ in segs], key, stream_offs=[so for _, _, so in segs],
                                     device=b.device, stream=st.handle)
        else:
            n = int(rng.integers(1, 96 << 20)) if rng.random() < 0.7 else int(rng.choice([65536 * 3, 65536 * 7 + 5, 131072 * 9 - 16, 1 << 20]))
            base = int(rng.integers(0, 4096))
            so = int(rng.choice([0, int(rng.integers(0, 1 << 62)), O.PERIOD - n // 3]))
            lo = max(0, base - 32)
            pt = rng.integers(0, 256, size=n + 64, dtype=np.uint8)
            b.upload(pt, offset=lo)
            for _ in range(passes):
                b.cycle(key, n=n, offset=base, stream_off=so, stream=st.handle)
            segs = [(base, n, so)]
        jobs.append((b, st, lo, key, pt, segs))
    for b, st, lo, key, pt, segs in jobs:
        st.sync()
        want = pt.copy()
        for o, n, so in segs:
            M.cycle_scalar_host(want[o - lo:o - lo + n], key, so)
        got = b.download(pt.size, offset=lo)
        if not np.array_equal(got, want):
            bad = np.flatnonzero(got != want)
            raise SystemExit(f"MISMATCH shape={shape} grid={grid} key={key:#x} segments={segs if len(segs) < 24 else len(segs)}: {bad.size} bytes differ, first at {bad[0]}")
        if n_cases % 16 == 0 and sum(n for _, n, _ in segs) <= (8 << 20):
            w2 = pt.copy()
            for o, n, so in segs:
                O.cycle_at(w2[o - lo:o - lo + n], key, so)
            assert np.array_equal(want, w2), "host loop vs oracle"
        n_cases += 1
        n_parts += len(segs)
        n_bytes += sum(n for _, n, _ in segs)
q = M.queue_stats()
print(f"SOAK_OK {n_cases} buffers ({n_parts} parts), {n_bytes / 1e9:.1f} GB cycled and compared in {seconds:.0f} s; work-queue bookkeeping {q}")
