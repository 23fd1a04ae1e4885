#!/usr/bin/env python3
"""tools/sweep_pcie_grid.py -- how many workgroups should a kernel that works ACROSS PCIe have, on how many lanes, and how should a
staged stream be cut once that is settled?  (the chunk kernels of the staged route; the cap, 256, was chosen in
round 2 on buffers of 64 MiB and up cycled by ONE kernel.)  Testing flavour, one process: the grid of over-PCIe launches, the
lanes and the chunking are set per row at run time.  Settings interleaved, best call of all repetitions kept (the box's CPU
share drifts).

    python3 tools/sweep_pcie_grid.py [grid|cut]
"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import modulate_amd as M  # noqa: E402

M.use_testing_flavour()
assert M.gpu_required() or True
what = sys.argv[1] if len(sys.argv) > 1 else "grid"
rng = np.random.default_rng(1)
PAGEABLE = (16, 64, 256)
PINNED = (1, 4, 16, 64, 128, 411)
bufs = {mib: rng.integers(0, 256, size=mib << 20, dtype=np.uint8) for mib in PAGEABLE}
pin = {}
for mib in PINNED:
    pb = M.PinnedBuffer((mib << 20) + 64)
    pb.array[:] = 5
    pin[mib] = pb


def best_of(fn, reps):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        b = min(b, time.perf_counter() - t0)
    return b


def apply(lanes=4, grid=0, split=16, chunk_min=2 << 20, ramp=1 << 20):
    M.debug_set_host_tunable("lanes", lanes)
    M.debug_set_host_tunable("split", split)
    M.debug_set_host_tunable("chunk_min_bytes", chunk_min)
    M.debug_set_host_tunable("ramp_bytes", ramp)
    M.debug_set_pcie_grid(grid)


res = {}
if what == "grid":
    settings = [(lanes, grid) for lanes in (2, 4, 6, 8) for grid in (0, 8, 16, 24, 32, 48, 64, 128, 256)]
    for rep in range(3):
        for lanes, grid in settings:
            apply(lanes=lanes, grid=grid)
            for mib, buf in bufs.items():
                k = (lanes, grid, "pageable", mib)
                res[k] = min(res.get(k, 1e9), best_of(lambda: M.cycle_host(buf, M.KEY_PS4), 5))
            if lanes == 4:
                for mib, pb in pin.items():
                    v = pb.array[4:4 + (mib << 20)]
                    k = (lanes, grid, "pinned", mib)
                    res[k] = min(res.get(k, 1e9), best_of(lambda: M.cycle_host(v, M.KEY_PS4), 5 if mib < 100 else 3))
    print("GB/s of payload, best call; grid = workgroups of an over-PCIe launch (0 = the product's rule: 32 up to 64 MiB, 256 beyond)")
    cols = [("pageable", m) for m in PAGEABLE]
    print("lanes  grid  " + "  ".join("%9s %3d MiB" % c for c in cols))
    for lanes, grid in settings:
        print("%5d  %4d  " % (lanes, grid) + "  ".join("%17.1f" % ((c[1] << 20) / res[(lanes, grid) + c] / 1e9) for c in cols))
    print("one kernel in place on page-locked memory (lanes do not matter):")
    cols = [("pinned", m) for m in PINNED]
    print("       grid  " + "  ".join("%9s %3d MiB" % c for c in cols))
    for grid in (0, 8, 16, 24, 32, 48, 64, 128, 256):
        print("       %4d  " % grid + "  ".join("%17.1f" % ((c[1] << 20) / res[(4, grid) + c] / 1e9) for c in cols))
else:
    settings = [(split, cmin, ramp, lanes) for split in (16, 32, 64) for cmin in (1, 2, 4) for ramp in (256, 512, 1024, 2048) for lanes in (4,)] + \
               [(16, 2, 1024, 2), (16, 2, 1024, 8), (32, 1, 512, 8), (32, 2, 512, 6)]
    for rep in range(3):
        for split, cmin, ramp, lanes in settings:
            apply(lanes=lanes, grid=0, split=split, chunk_min=cmin << 20, ramp=ramp << 10)
            for mib, buf in bufs.items():
                k = (split, cmin, ramp, lanes, mib)
                res[k] = min(res.get(k, 1e9), best_of(lambda: M.cycle_host(buf, M.KEY_PS4), 5))
    print("GB/s of payload, best call, pageable; the product's grid rule")
    print("split  chunk_min_MiB  ramp_KiB  lanes  " + "  ".join("%3d MiB" % m for m in PAGEABLE))
    for split, cmin, ramp, lanes in settings:
        print("%5d  %13d  %8d  %5d  " % (split, cmin, ramp, lanes) + "  ".join("%7.1f" % ((m << 20) / res[(split, cmin, ramp, lanes, m)] / 1e9) for m in PAGEABLE))
