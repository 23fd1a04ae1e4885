#!/usr/bin/env python3
"""tools/archive/first_launch_where.py [reps=4] -- what makes a process's first 411 MB launch 8 us slower in bench.py's preamble than in a process that
owns only a 411 MB part (profiles/r05_first_launch.txt, run E)?  Each line below is a FRESH process (the first launch happens once per process)
that allocates, uploads as bench.py does (64 MiB tiles from pageable memory), then times ONE 411 MB launch and the next one with HIP events:

  alloc411_up411         a 411 MB allocation, uploaded once                                  (run E, line 1)
  alloc4G_up4G_head      a 4 GiB allocation, all of it uploaded, its first 411 MB cycled     (run E, line 2 = bench.py)
  alloc4G_up411_head     a 4 GiB allocation, only its first 411 MB uploaded                  -> the allocation's size alone
  alloc411_up411x10      a 411 MB allocation, uploaded ten times over (as long as 4 GiB takes) -> the upload's length alone
  alloc4G_up4G_tail      a 4 GiB allocation, all uploaded, its LAST 411 MB cycled (what the upload touched last)
  alloc4G_up4G_head_idle as alloc4G_up4G_head, 50 ms of idleness between upload and launch
  ..._onecall            the same bytes by ONE modgpu_h2d call per pass over the buffer instead of one per 64 MiB tile
  ..._raw                the tiles by the runtime's own hipMemcpy: no empty launch when a copy starts (what modgpu_h2d was until round 5)
  ..._rawlast            hipMemcpy for every tile but the last, which goes through modgpu_h2d: ONE empty launch, 1.4 ms before the upload ends
  ..._sleepfirst         1.5 s of idleness between the allocation (whose device preparation launches kernels) and the upload, as tools/first_pass has

Not part of the test-suite; prints a table."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ("alloc411_up411", "alloc4G_up4G_head", "alloc4G_up411_head", "alloc411_up411x10", "alloc4G_up4G_tail", "alloc4G_up4G_head_idle",
         "alloc4G_up4G_head_onecall", "alloc411_up411x10_onecall", "alloc4G_up4G_head_raw", "alloc4G_up4G_head_rawlast", "alloc4G_up4G_head_sleepfirst")
if os.environ.get("WHERE_ONLY"):
    CASES = tuple(c for c in CASES if c in os.environ["WHERE_ONLY"].split(","))
SMALL = 411 * 1000 * 1000


def child(case):
    import time

    import numpy as np
    sys.path.insert(0, ROOT)
    import modulate_amd as M
    big = 1 << 32
    alloc = SMALL if case.startswith("alloc411") else big
    up = SMALL if "up411" in case else big
    times = 10 if "x10" in case else 1
    tile = np.random.default_rng(1).integers(0, 256, size=64 << 20, dtype=np.uint8)
    if case.endswith("_onecall"):
        tile = np.resize(tile, up)
    part = M.DeviceBuffer(alloc, device=0)
    raw = None
    if "_raw" in case:
        import ctypes
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hip_rt
        raw = hip_rt.hip()
    if case.endswith("_sleepfirst"):
        time.sleep(1.5)
    t0 = time.perf_counter()
    for _ in range(times):
        for off in range(0, up, tile.size):
            l = min(tile.size, up - off)
            if raw is not None and not (case.endswith("_rawlast") and off + l >= up):
                rc = raw.hipMemcpy(ctypes.c_void_p(part.ptr + off), ctypes.c_void_p(tile.ctypes.data), ctypes.c_size_t(l), 1)
                assert rc == 0, rc
            else:
                part.upload(tile[:l], offset=off)
    part.sync()
    t_up = time.perf_counter() - t0
    if case.endswith("_idle"):
        time.sleep(0.05)
    at = (alloc - SMALL) // 4096 * 4096 if case.endswith("_tail") else 0
    first = M.time_cycle_device(part.ptr + at, SMALL, M.KEY_PS4, 0, 0, None, iters=1)
    nxt = M.time_cycle_device(part.ptr + at, SMALL, M.KEY_PS4, 0, 0, None, iters=1)
    third = M.time_cycle_device(part.ptr + at, SMALL, M.KEY_PS4, 0, 0, None, iters=8)
    print(f"{first:.4f} {nxt:.4f} {third:.4f} {t_up:.2f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    print(f"{'case':24s} first launch (ms) | next | mean of the following 8 | upload (s)      [{reps} fresh processes each, interleaved]")
    rows = {c: [] for c in CASES}
    for _ in range(reps):
        for c in CASES:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", c], capture_output=True, text=True, timeout=120)
            if out.returncode != 0:
                raise SystemExit(f"{c}: {out.stderr[-400:]}")
            rows[c].append(out.stdout.split())
    for c in CASES:
        cols = list(zip(*rows[c]))
        print(f"{c:24s} first {' '.join(cols[0])} | next {' '.join(cols[1])} | then {' '.join(cols[2])} | upload {' '.join(cols[3])}", flush=True)
