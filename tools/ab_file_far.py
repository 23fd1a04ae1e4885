#!/usr/bin/env python3
"""tools/ab_file_far.py [reps=5] [MiB,...] -- a part file whose cached pages sit on the OTHER socket than the GPU's, read into memory
(modgpu_cycle_file_to_host): MODGPU_NUMA=1 (the staging set of the file's node: slots and workers there, only the GPU crosses the
socket link) against MODGPU_NUMA=0 (the set next to the GPU: pread crosses).  A child process per setting (the variable is read at
load), interleaved; each child moves itself to the other node's CPUs before it writes the file on tmpfs.  GB/s of payload, best / median."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time, json, numpy as np
sys.path.insert(0, %r)
import modulate_amd as M
mib, reps, where, gpu_node = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
def cpus(n):
    out = []
    for part in open("/sys/devices/system/node/node%%d/cpulist" %% n).read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out
nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
pick = [k for k in nodes if (k == gpu_node) == (where == "near") and set(cpus(k)) & os.sched_getaffinity(0)]
if gpu_node < 0 or not pick:
    print(json.dumps({"skip": True})); sys.exit(0)
os.sched_setaffinity(0, set(cpus(pick[0])) & os.sched_getaffinity(0))
n = mib << 20
path = "/dev/shm/ab_file_far_%%d.part" %% os.getpid()
np.resize(np.random.default_rng(3).integers(0, 256, size=min(n, 1 << 26), dtype=np.uint8), n).tofile(path)
res = {}
pb = M.PinnedBuffer(n + 64)
for name, dst in (("pageable", np.zeros(n, np.uint8)), ("page_locked", pb.array[4:4 + n])):
    ts = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        M.cycle_file_to_host(path, n, M.KEY_PS4, out=dst)
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[1:])
    res[name] = [round(n / ts[0] / 1e9, 2), round(n / ts[len(ts) // 2] / 1e9, 2)]
os.unlink(path)
res["other_nodes_set_calls"] = M.host_pool_stats()["calls_on_another_nodes_set"]
print(json.dumps(res))
"""
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 392]
sys.path.insert(0, ROOT)
import modulate_amd as M  # noqa: E402
GPU_NODE = M.device_numa_node(0)  # (asked here: a child under MODGPU_NUMA=0 is not told)
print("the GPU hangs off NUMA node", GPU_NODE)
print("file (tmpfs) -> memory, the file written from the socket named; MODGPU_NUMA=0: always the staging set next to the GPU; GB/s of payload best / median")
for mib in sizes:
    for where in ("far", "near"):
        for numa in ("1", "0", "1", "0"):
            r = subprocess.run([sys.executable, "-c", CHILD % ROOT, str(mib), str(reps), where, str(GPU_NODE)], capture_output=True, text=True, env=dict(os.environ, MODGPU_NUMA=numa), timeout=600)
            out = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else r.stderr[-300:]
            print(f"  {mib:5d} MiB  file on the {where} socket  MODGPU_NUMA={numa}  {out}", flush=True)
