"""Child process of tests/test_gpu_parity.py::test_first_host_buffer_calls_of_a_process_on_eight_devices_at_once: the FIRST host-buffer
calls of a fresh process, eight at once (eight logical devices aliased onto the one GPU, MODGPU_DEVICE_ALIAS=8): every staging context,
slot, flag word and counter is made while seven other threads do the same.  Round 6 found the host-fed kernel's counters cleared on the
NULL stream and the kernel launched -- on its own non-blocking stream -- before the fill had run (13 of 25 fresh processes returned wrong
bytes).  Prints FIRST_CALL_OK, or where the bytes differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import modulate_amd as M  # noqa: E402
from oracle import oracle as O  # noqa: E402

assert M.device_count() == 8 and M.gpu_required()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 96
sizes = [(mib << 20) + 17 * i for i in range(8)]
base = [O.splitmix_bytes(s, 0x4D6F64756C617465 + i) for i, s in enumerate(sizes)]
want = [O.cycle(k.copy(), O.KEY_PS4) for k in base]
parts = [k.copy() for k in base]
M.cycle_parts_host(parts, M.KEY_PS4, 8)  # pageable memory: the host-fed kernel, one per device, all for the first time
bad = 0
for i, (p, w) in enumerate(zip(parts, want)):
    d = np.flatnonzero(p != w)
    if d.size:
        bad += 1
        print(f"part {i}: {d.size} bytes differ, first at {int(d[0])} (chunk {int(d[0]) // 262144}, +{int(d[0]) % 262144}); still plaintext: {int((p[d] == base[i][d]).sum())}")
assert M.last_launch()["variant"] == 4 and M.path_stats()["scalar_calls"] == 0
print("FIRST_CALL_OK" if not bad else f"FIRST_CALL_MISMATCH in {bad} parts")
sys.exit(1 if bad else 0)
