#!/usr/bin/env python3
"""tools/archive/where_pinned_lands.py -- on which NUMA node does the runtime put page-locked memory (hipHostMalloc), seen from a thread on each node?
The staging set "next to the GPU" takes its slots from hipHostMalloc and binds its workers to the node sysfs names for the GPU: if the two
disagree every staging copy crosses the socket link.  Prints the node of the first / middle / last page of an 8 MiB allocation."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import modulate_amd as M  # noqa: E402
import hip_rt  # noqa: E402

libc = ctypes.CDLL(None, use_errno=True)


def node_of(addr):
    node = ctypes.c_int(-1)
    rc = libc.syscall(239, ctypes.byref(node), None, ctypes.c_ulong(0), ctypes.c_void_p(addr), ctypes.c_ulong(3))  # get_mempolicy, MPOL_F_NODE | MPOL_F_ADDR
    return node.value if rc == 0 else -1


def cpus_of(n):
    out = []
    for part in open(f"/sys/devices/system/node/node{n}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return set(out)


assert M.device_count() >= 1
M.DeviceBuffer(4096, device=0)  # the runtime is up
hip = hip_rt.hip()
bdf = ctypes.create_string_buffer(64)
hip.hipDeviceGetPCIBusId(bdf, 63, 0)
gpu_node = open(f"/sys/bus/pci/devices/{bdf.value.decode().lower()}/numa_node").read().strip()
nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
print(f"GPU 0 at {bdf.value.decode()}: sysfs numa_node = {gpu_node}; nodes {nodes}")
allowed = os.sched_getaffinity(0)
for n in nodes:
    cp = cpus_of(n) & allowed
    if not cp:
        continue
    os.sched_setaffinity(0, cp)
    for flags, name in ((0, "hipHostMallocDefault"), (0x1 | 0x2, "Portable|Mapped"), (0x1 | 0x2 | 0x20000000, "Portable|Mapped|NumaUser")):
        p = ctypes.c_void_p()
        rc = hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(8 << 20), ctypes.c_uint(flags))
        if rc != 0:
            print(f"  thread on node {n}: {name}: hip error {rc}")
            continue
        where = [node_of(p.value + o) for o in (0, 4 << 20, (8 << 20) - 4096)]
        print(f"  thread on node {n} (cpu {libc.sched_getcpu()}): {name:28s} pages on nodes {where}")
        hip.hipHostFree(p)
os.sched_setaffinity(0, allowed)
