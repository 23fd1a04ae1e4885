#!/usr/bin/env python3
"""tools/archive/ab_feed_processes.py [processes=8] [MiB=64] -- is a process's rate on the pageable route a property of the PROCESS (where the scheduler and the
allocator put it) or of the schedule?  Each line is a fresh process that runs the launch-per-chunk schedule and the host-fed kernel interleaved on
one pageable buffer (testing flavour, 12 calls each) and says where its thread and pages were."""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(mib):
    import numpy as np
    sys.path.insert(0, ROOT)
    import modulate_amd as M
    M.use_testing_flavour()
    libc = ctypes.CDLL(None)
    n = mib << 20
    buf = np.random.default_rng(1).integers(0, 256, size=n, dtype=np.uint8)
    node = ctypes.c_int(-1)
    libc.syscall(239, ctypes.byref(node), None, ctypes.c_ulong(0), ctypes.c_void_p(buf.ctypes.data + n // 2), ctypes.c_ulong(3))
    t = {0: [], 1: []}
    for rep in range(14):
        for feed in (0, 1):
            M.debug_set_host_tunable("feed", feed)
            t0 = time.perf_counter()
            M.cycle_host(buf, M.KEY_PS4)
            if rep >= 2:
                t[feed].append(time.perf_counter() - t0)
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    print(f"cpu {libc.sched_getcpu():3d}  pages on node {node.value}  other-node-set calls {M.host_pool_stats()['calls_on_another_nodes_set']:3d}   "
          f"launch per chunk {n / min(t[0]) / 1e9:5.2f} / {n / med[0] / 1e9:5.2f}   host-fed {n / min(t[1]) / 1e9:5.2f} / {n / med[1] / 1e9:5.2f}  GB/s best / median")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
        sys.exit(0)
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    for _ in range(procs):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(mib)], capture_output=True, text=True, timeout=120)
        print(out.stdout.strip() or out.stderr[-300:], flush=True)
