#!/usr/bin/env python3
"""Host-buffer rate of modgpu_cycle_host on page-locked caller memory (modgpu_host_alloc), by route:
   dma     H2D from the caller's pages -> kernel in HBM -> D2H back, ring of R device slots of C MiB
   kernel  one launch over PCIe on the pages themselves, by launch shape and grid
and, beside them, the staged route on pageable memory (the r01 path).  Ring / chunk are read once at
library load, so each (R, C) runs in a child process.  PCIe-inclusive payload GB/s (each byte crosses twice)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SIZES = (64 << 20, 411 << 20, 1 << 30, 1 << 32)
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import modulate_amd as M
M.use_testing_flavour()  # the route / shape selectors exist only in libmodgpu_testing.so
what = sys.argv[1]
def timeit(fn, n):
    fn()
    reps = 5 if n < (1 << 30) else 3
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t0) / reps
    return dt
for n in %r:
    if what.startswith("pageable"):
        M.debug_set_staged_mode(2 if what == "pageable_slot_kernel" else 1)
        buf = np.empty(n, np.uint8); buf[:] = 7
        dt = timeit(lambda: M.cycle_host(buf, M.KEY_PS4), n)
        print(f"   {what:24s} n={n>>20:5d} MiB  {dt*1e3:8.2f} ms  {n/dt/1e9:6.1f} GB/s", flush=True)
        continue
    pb = M.PinnedBuffer(n + 64); pb.array[:] = 7
    view = pb.array[4:4 + n]            # buf+4, like the reference's callers
    if what == "dma":
        M.debug_set_pinned_mode(1)
        dt = timeit(lambda: M.cycle_host(view, M.KEY_PS4), n)
        print(f"   pinned dma               n={n>>20:5d} MiB  {dt*1e3:8.2f} ms  {n/dt/1e9:6.1f} GB/s", flush=True)
    else:
        M.debug_set_pinned_mode(2)
        for shape in ("small", "large"):
            for grid in (16, 32, 64, 128, 256, 0):
                if shape == "small" and grid == 0 and n > (1 << 30): continue
                M.debug_set_launch(shape, grid)
                dt = timeit(lambda: M.cycle_host(view, M.KEY_PS4), n)
                print(f"   pinned kernel {shape:5s} grid<={grid:4d} n={n>>20:5d} MiB  {dt*1e3:8.2f} ms  {n/dt/1e9:6.1f} GB/s", flush=True)
        M.debug_set_launch(None, 0)
    M.debug_set_pinned_mode(0)
    pb.free()
''' % (ROOT, SIZES)


def run(what, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    print(f"== {what} {env}", flush=True)
    subprocess.run([sys.executable, "-c", CHILD, what], env=e, check=False)


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    if len(sys.argv) > 1 and sys.argv[1] == "staged":  # pageable memory: DMA per chunk vs kernel on the pinned slot
        for pipes in (4, 6, 8, 12):
            for chunk in (4, 8, 16):
                run("pageable_dma", MODGPU_HOST_PIPES=pipes, MODGPU_HOST_CHUNK_MB=chunk)
                run("pageable_slot_kernel", MODGPU_HOST_PIPES=pipes, MODGPU_HOST_CHUNK_MB=chunk)
        sys.exit(0)
    run("pageable_dma")
    for ring in ((4,) if quick else (2, 3, 4)):
        for chunk in ((16,) if quick else (4, 8, 16, 32, 64)):
            run("dma", MODGPU_HOST_RING=ring, MODGPU_HOST_CHUNK_MB=chunk)
    run("kernel")
