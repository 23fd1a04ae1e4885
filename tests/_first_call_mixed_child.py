"""Child process of tests/test_gpu_parity.py::test_first_calls_of_a_process_of_every_kind_at_once_on_one_device: a fresh process whose FIRST
host-buffer calls are six at once on ONE device, one of each kind -- pageable memory (host-fed kernel), page-locked memory in place, a part
file into pageable and into page-locked memory, a header-sized buffer, a small pageable buffer -- so that the staging context, its
slots, worker threads, flag words and counters are all made while other routes are being set up beside them.  Prints MIXED_FIRST_OK."""
import os
import sys
import tempfile
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import modulate_amd as M  # noqa: E402
from oracle import oracle as O  # noqa: E402

assert M.device_count() >= 1 and M.gpu_required()
n = (24 << 20) + 11
pt = O.splitmix_bytes(n + 64, 77)
want = pt[5:5 + n].copy()
O.cycle_at(want, O.KEY_PS4, 0)
d = tempfile.mkdtemp(prefix="first_mixed_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "part.ark")
pt[5:5 + n].tofile(path)
go = threading.Barrier(6)
errors = []


def run(name, fn):
    try:
        go.wait()
        fn()
    except Exception as e:  # noqa: BLE001
        errors.append((name, repr(e)))


def pageable():
    buf = pt.copy()
    M.cycle_host(buf[5:5 + n], M.KEY_PS4)
    assert np.array_equal(buf[5:5 + n], want) and np.array_equal(buf[:5], pt[:5]) and np.array_equal(buf[5 + n:], pt[5 + n:])


def locked_in_place():
    pb = M.PinnedBuffer(n + 64)
    pb.array[:] = pt
    M.cycle_host(pb.array[5:5 + n], M.KEY_PS4)
    assert np.array_equal(pb.array[5:5 + n], want) and np.array_equal(pb.array[:5], pt[:5])
    pb.free()


def file_to_pageable():
    assert np.array_equal(M.cycle_file_to_host(path, n, M.KEY_PS4), want)


def file_to_locked():
    pb = M.PinnedBuffer(n + 64)
    pb.array[:] = 0xEE
    M.cycle_file_to_host(path, n, M.KEY_PS4, out=pb.array[7:7 + n])
    assert np.array_equal(pb.array[7:7 + n], want) and np.all(pb.array[:7] == 0xEE) and np.all(pb.array[7 + n:] == 0xEE)
    pb.free()


def header_sized():
    b = pt[:300_000].copy()
    w = b.copy()
    O.cycle_at(w, O.KEY_PS3, 9)
    M.cycle_host(b, M.KEY_PS3, stream_off=9)
    assert np.array_equal(b, w)


def small_pageable():
    b = pt[:(3 << 20) + 1].copy()
    w = b.copy()
    O.cycle_at(w, O.KEY_PS4, 0)
    M.cycle_host(b, M.KEY_PS4)
    assert np.array_equal(b, w)


threads = [threading.Thread(target=run, args=(f.__name__, f)) for f in (pageable, locked_in_place, file_to_pageable, file_to_locked, header_sized, small_pageable)]
for t in threads:
    t.start()
for t in threads:
    t.join()
os.unlink(path)
os.rmdir(d)
assert M.path_stats()["scalar_calls"] == 0
print("MIXED_FIRST_OK" if not errors else f"MIXED_FIRST_FAILED {errors}")
sys.exit(1 if errors else 0)
