#!/usr/bin/env python3
"""tools/soak_host.py [seconds=120] [seed=1] -- randomized long-run check of the HOST-buffer routes (not part of the test-suite).

What CEncryptionCycler::Cycle and the CArk part cipher run on: modgpu_cycle_auto_host / modgpu_cycle_host / the file routes over
pageable, page-locked and registered memory of 1 B ... 160 MiB, any misalignment, stream offsets and keys, one to three callers at
once on one GPU, the calling thread moved between the NUMA nodes (so that both staging sets of the device serve), and -- in a
quarter of the cases -- a failure injected at a random piece and stage of the call, which the library has to survive (the host
loop finishes the pieces that have not arrived) unless the buffer is page-locked and the kernel dies under way (the one documented
error).  Every result is compared whole against the library's own host loop on a copy (a different engine from the one under test,
itself pinned to the oracle and the golden vectors by tests/test_capi_cpu.py), every 8th also against the oracle.
Prints a summary line; any mismatch raises."""
import os
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.pop("MODGPU_REQUIRE_GPU", None)  # the host loop is both the checker and the second engine here
os.environ["MODGPU_MIN_GPU_BYTES"] = "65536"  # every buffer of 64 KiB and more is the kernel's
import modulate_amd as M  # noqa: E402
from oracle import oracle as O  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
M.use_testing_flavour()
assert M.device_count() >= 1 and not M.gpu_required()
KEYS = (M.KEY_PS4, M.KEY_PS3, 1, 0xFFFFFFFF, 12345, 0x7FFFFFFF)
STAGES = (M.STAGE_FILL, M.STAGE_LAUNCH, M.STAGE_SYNC, M.STAGE_DRAIN, M.STAGE_AFTER_DRAIN)


def node_cpus():
    out = {}
    base = "/sys/devices/system/node"
    for d in os.listdir(base) if os.path.isdir(base) else ():
        if d.startswith("node") and d[4:].isdigit():
            cpus = []
            for part in open(f"{base}/{d}/cpulist").read().strip().split(","):
                a, _, b = part.partition("-")
                cpus += list(range(int(a), int(b or a) + 1))
            cpus = set(cpus) & os.sched_getaffinity(0)
            if cpus:
                out[int(d[4:])] = cpus
    return out


NODES = node_cpus()
ALL = os.sched_getaffinity(0)
tile = O.splitmix_bytes(1 << 22, seed)
tmpdir = tempfile.mkdtemp(prefix="soak_host_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
counts = {"calls": 0, "bytes": 0, "injected": 0, "rescued": 0, "errors_as_documented": 0, "threads": 0, "files": 0, "oracle_checks": 0}
lock = threading.Lock()


def expected(pt, key, off):
    w = pt.copy()
    M.cycle_scalar_host(w, key, stream_off=off)
    return w


def one_case(r, inject_ok):
    n = int(2 ** r.uniform(0, 27.3))  # 1 B .. ~160 MiB, log-uniform
    if r.random() < 0.15:
        n = int(r.choice([4092, 65536, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, (8 << 20) + 5, (64 << 20) - 3]))
    key = int(KEYS[int(r.integers(0, len(KEYS)))])
    off = int(r.choice([0, 0, 7, O.PERIOD - 1000, (1 << 33) + 5]))
    mis = int(r.integers(0, 64))
    kind = r.choice(["pageable", "pageable", "pinned", "file_to_host", "host_to_file"]) if n >= 4096 else "pageable"
    pt = np.resize(np.roll(tile, int(r.integers(0, tile.size))), n)
    want = expected(pt, key, off)
    inject = inject_ok and n > (1 << 20) and r.random() < 0.25
    piece, stage = (int(r.choice([0, M.INJECT_PIECE_MIDDLE, M.INJECT_PIECE_LAST, int(r.integers(0, 40))])), int(r.choice(STAGES))) if inject else (0, -1)
    doc_error = False
    if kind == "pageable":
        hold = np.empty(n + mis + 64, np.uint8)
        hold[:] = 0xEE
        buf = hold[mis:mis + n]
        buf[:] = pt
        if inject:
            M.debug_inject_failure_at(piece, stage)
        M.cycle_auto_host(buf, key, stream_off=off)
        assert (hold[:mis] == 0xEE).all() and (hold[mis + n:] == 0xEE).all(), "guard bytes"
        got = buf
    elif kind == "pinned":
        pb = M.PinnedBuffer(n + mis + 64)
        pb.array[:] = 0xEE
        buf = pb.array[mis:mis + n]
        buf[:] = pt
        if inject:
            M.debug_inject_failure_at(0, stage)
        try:
            M.cycle_auto_host(buf, key, stream_off=off)
            got = buf.copy()
            assert (pb.array[:mis] == 0xEE).all() and (pb.array[mis + n:mis + n + 64] == 0xEE).all(), "guard bytes"
        except M.ModGpuError as e:  # the kernel "died" while it worked in place on page-locked memory: the one case that stays an error
            assert inject and stage in (M.STAGE_SYNC,) and e.code == 3 and "page-locked" in str(e), (stage, str(e))
            doc_error, got = True, want
        pb.free()
    else:
        path = os.path.join(tmpdir, f"f{threading.get_ident()}")
        with lock:
            counts["files"] += 1
        if kind == "file_to_host":
            pt.tofile(path)
            dst = np.empty(n, np.uint8)
            if inject:
                M.debug_inject_failure_at(piece, stage)
            M.cycle_file_to_host(path, n, key, stream_off=off, out=dst)
            got = dst
        else:
            if inject:
                M.debug_inject_failure_at(piece, stage)
            M.cycle_host_to_file(pt, path, key, stream_off=off)
            got = np.fromfile(path, dtype=np.uint8)
        os.unlink(path)
    if inject and M.debug_injection_armed():  # (the call had no such piece / stage, e.g. a one-piece call asked for DRAIN: disarm)
        M.debug_inject_failure_at(0, -1)
        inject = False
    assert np.array_equal(got, want), (kind, n, mis, off, hex(key), inject, piece, stage, int(np.flatnonzero(got != want)[0]) if got.size == want.size else "size")
    with lock:
        counts["calls"] += 1
        counts["bytes"] += n
        counts["injected"] += 1 if inject else 0
        counts["errors_as_documented"] += 1 if doc_error else 0
        if counts["calls"] % 8 == 0 and n <= (32 << 20):
            w = pt.copy()
            O.cycle_at(w, key, off)
            assert np.array_equal(w, want), "the checker disagrees with the oracle"
            counts["oracle_checks"] += 1


t_end = time.time() + seconds
t_note = time.time() + 60
before = M.path_stats()
while time.time() < t_end:
    if time.time() > t_note:
        print(f"... {counts['calls']} calls, {counts['bytes'] / 1e9:.1f} GB so far", flush=True)
        t_note = time.time() + 60
    if len(NODES) > 1 and rng.random() < 0.3:  # this thread -- and the pages it touches from now on -- on one node or the other
        os.sched_setaffinity(0, NODES[int(rng.choice(sorted(NODES)))])
    elif rng.random() < 0.1:
        os.sched_setaffinity(0, ALL)
    if rng.random() < 0.2:  # several callers at once on the one GPU (no injection: the armed failure is process-wide)
        k = int(rng.integers(2, 4))
        errs = []

        def run(i, s):
            try:
                one_case(np.random.default_rng(s), False)
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))
        ts = [threading.Thread(target=run, args=(i, int(rng.integers(0, 1 << 62)))) for i in range(k)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
        counts["threads"] += k
    else:
        one_case(rng, True)
os.sched_setaffinity(0, ALL)
after = M.path_stats()
counts["rescued"] = after["midcall_rescues"] - before["midcall_rescues"]
pool = M.host_pool_stats()
os.rmdir(tmpdir)
print(f"SOAK_HOST_OK {counts['calls']} calls ({counts['threads']} of them from concurrent callers, {counts['files']} file routes), {counts['bytes'] / 1e9:.1f} GB cycled and compared in {seconds:.0f} s; "
      f"{counts['injected']} failures injected mid-call, {counts['rescued']} calls finished by the host loop ({after['midcall_rescued_bytes'] / 1e9:.2f} GB), "
      f"{counts['errors_as_documented']} page-locked in-place errors as documented; {pool['calls_on_another_nodes_set']} calls on another NUMA node's staging set; "
      f"{counts['oracle_checks']} results also checked against the oracle; path stats {after}")
