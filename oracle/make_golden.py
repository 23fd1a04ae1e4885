#!/usr/bin/env python3
"""Generate tests/golden/cycle_golden.json from the COMPILED REFERENCE (oracle/_ref).

Run in the build container only (it needs oracle/_ref/libref_cycler.so, built by
`make -C oracle ref` from /root/reference/Modulate/CEncryptionCycler.cpp).  The output is
data only: keys, offsets, keystream bytes (hex) and FNV-1a-64 digests.  The GPU box uses the
committed JSON; it never sees the reference.

    python oracle/make_golden.py [--big]     # --big adds the 2^32-1 byte run (~35 s, 4 GiB RAM)
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402

KEYS = [0xC64EED30, 0x90CFC0AB, 0, 0x7FFFFFFF, 0x80000001, 1, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFE,
        12345, (-127772) & 0xFFFFFFFF, 127773, 0x7FFFFFFD, 2, 16807, 0xDEADBEEF]
P = O.PERIOD
SEED = 0x4D6F64756C617465


def ref_ks(key, n):
    return O.ref_cycle(np.zeros(n, dtype=np.uint8), key)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cycle_golden.json"))
    a = ap.parse_args()
    assert O.have_ref(), "build oracle/_ref first: make -C oracle ref"
    g = {"generator": "oracle/make_golden.py over oracle/_ref (reference CEncryptionCycler.cpp, g++ -O2)",
         "fnv": "FNV-1a 64, offset basis cbf29ce484222325",
         "keystream": [], "cycle_key": [], "plaintext_cases": [], "large": None}
    for key in KEYS:
        ks = ref_ks(key, 1 << 20)
        g["keystream"].append({
            "key": key, "first64": ks[:64].tobytes().hex(),
            "fnv_4k": f"{O.fnv1a64(ks[:4096]):016x}",
            "fnv_1m": f"{O.fnv1a64(ks):016x}",
            "at_1m_minus_16": ks[-16:].tobytes().hex(),
        })
    # CycleKey is private in the reference; observe it through Cycle (4-step keystreams).
    for key in KEYS + [0x12345678, 0x7FFFFFFF - 1, 0x41A7, 0x1F31D, 0x1F31C, 0xFFFE0CE4]:
        g["cycle_key"].append({"key": key, "ks4": ref_ks(key, 4).tobytes().hex()})
    # plaintext cases (sizes around the kernel's word/tile edges), PS3 + PS4 keys
    for key in (0x90CFC0AB, 0xC64EED30):
        for n in (0, 1, 15, 16, 17, 63, 64, 65, 255, 1023, 1024, 1025, 4092, 4096, 65537, (1 << 20) - 1, (1 << 20) + 1):
            pt = O.splitmix_bytes(n, SEED + n)
            ct = O.ref_cycle(pt.copy(), key)
            g["plaintext_cases"].append({"key": key, "n": n, "seed": SEED + n,
                                         "pt_fnv": f"{O.fnv1a64(pt):016x}", "ct_fnv": f"{O.fnv1a64(ct):016x}",
                                         "ct_first16": ct[:16].tobytes().hex(), "ct_last16": ct[-16:].tobytes().hex()})
    # b[i] = (i*131+7)&0xFF, 4096 B, PS4 key
    b = ((np.arange(4096, dtype=np.uint32) * 131 + 7) & 0xFF).astype(np.uint8)
    ct = O.ref_cycle(b.copy(), 0x90CFC0AB)
    g["survey_4k"] = {"key": 0x90CFC0AB, "ct_fnv": f"{O.fnv1a64(ct):016x}"}
    if a.big:
        n = (1 << 32) - 1
        ks = ref_ks(0x90CFC0AB, n)
        g["large"] = {
            "key": 0x90CFC0AB, "n": n,
            "around_period": {"start": P - 8, "hex": ks[P - 8:P + 16].tobytes().hex()},
            "tail16": {"start": n - 16, "hex": ks[n - 16:].tobytes().hex()},
            "fnv_4k": f"{O.fnv1a64(ks[:4096]):016x}",
            "fnv_at_2g_1m": f"{O.fnv1a64(ks[1 << 31:(1 << 31) + (1 << 20)]):016x}",
            "fnv_all": f"{O.fnv1a64(ks):016x}",
            "period_repeats_1g": bool(np.array_equal(ks[:1 << 30], ks[P:P + (1 << 30)])),
            "samples": [{"off": int(o), "hex": ks[int(o):int(o) + 64].tobytes().hex()}
                        for o in (1 << 24, (1 << 28) + 5, (1 << 30) - 7, (1 << 31) - 64, (1 << 31) + 12345, 3 * (1 << 30) + 1, n - 64)],
        }
        del ks
    else:
        try:
            with open(a.out) as f:
                g["large"] = json.load(f).get("large")
        except OSError:
            pass
    with open(a.out, "w") as f:
        json.dump(g, f, indent=1)
    print("wrote", os.path.normpath(a.out))


if __name__ == "__main__":
    main()
