"""ctypes bindings over the CPU checker libraries (TEST INFRASTRUCTURE, not product).

* ``liboracle_cycle.so`` -- our C restatement of Modulate/CEncryptionCycler.cpp:4-25
  (oracle/cycle_oracle.c).
* ``_ref/libref_cycler.so`` -- the reference's own CEncryptionCycler.cpp compiled where it
  lies (oracle/Makefile target ``ref``); optional, absent if it was never built.

Nothing here reads /root/reference at run time.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "liboracle_cycle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libref_cycler.so")

MAGIC_PS3 = 0xC64EED30  # Settings.h:16
MAGIC_PS4 = 0x6F303F55  # Settings.h:17
KEY_PS3 = 0xC64EED30  # Settings.h:19
KEY_PS4 = 0x90CFC0AB  # Settings.h:20
LCG_M = 0x7FFFFFFF
LCG_A = 16807
PERIOD = LCG_M - 1
FNV_OFFSET = 0xCBF29CE484222325

__all__ = [
    "MAGIC_PS3", "MAGIC_PS4", "KEY_PS3", "KEY_PS4", "LCG_M", "LCG_A", "PERIOD", "FNV_OFFSET",
    "build", "as_int32", "cycle_key", "cycle", "cycle_serial64", "cycle_at", "state_at",
    "keystream_at", "keystream", "fnv1a64", "hdr_decrypt", "hdr_encrypt", "have_ref", "ref_cycle",
    "pure_cycle_key", "pure_cycle", "splitmix_bytes",
]


def build(ref=True):
    """(Re)build the checker libraries with gcc.  `ref` also tries the compiled reference."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def as_int32(key):
    """The reference passes `const unsigned int` keys into an `int` parameter (CArk.cpp:336-339)."""
    key &= 0xFFFFFFFF
    return key - (1 << 32) if key & 0x80000000 else key


_lib = None


def _oracle():
    global _lib
    if _lib is None:
        if not os.path.exists(_ORACLE_SO):
            build(ref=False)
        lib = ctypes.CDLL(_ORACLE_SO)
        u8p = ctypes.c_void_p
        lib.oracle_cycle_key.restype = ctypes.c_int32
        lib.oracle_cycle_key.argtypes = [ctypes.c_int32]
        lib.oracle_cycle.restype = None
        lib.oracle_cycle.argtypes = [u8p, ctypes.c_uint32, ctypes.c_int32]
        lib.oracle_cycle_serial64.restype = None
        lib.oracle_cycle_serial64.argtypes = [u8p, ctypes.c_uint64, ctypes.c_int32]
        lib.oracle_state_at.restype = ctypes.c_int32
        lib.oracle_state_at.argtypes = [ctypes.c_int32, ctypes.c_uint64]
        lib.oracle_keystream_at.restype = ctypes.c_uint8
        lib.oracle_keystream_at.argtypes = [ctypes.c_int32, ctypes.c_uint64]
        lib.oracle_cycle_at.restype = None
        lib.oracle_cycle_at.argtypes = [u8p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint64]
        lib.oracle_fnv1a64.restype = ctypes.c_uint64
        lib.oracle_fnv1a64.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint64]
        lib.oracle_hdr_decrypt.restype = ctypes.c_int
        lib.oracle_hdr_decrypt.argtypes = [u8p, ctypes.c_uint32]
        lib.oracle_hdr_encrypt.restype = ctypes.c_int
        lib.oracle_hdr_encrypt.argtypes = [u8p, ctypes.c_uint32, ctypes.c_int]
        _lib = lib
    return _lib


def _ptr(a):
    assert isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"] and a.flags["WRITEABLE"]
    return ctypes.c_void_p(a.ctypes.data)


def cycle_key(key):
    return _oracle().oracle_cycle_key(as_int32(key))


def cycle(buf, key):
    """In-place reference Cycle over a uint8 ndarray (n <= 2^32-1)."""
    assert buf.size <= 0xFFFFFFFF
    _oracle().oracle_cycle(_ptr(buf), buf.size, as_int32(key))
    return buf


def cycle_serial64(buf, key):
    _oracle().oracle_cycle_serial64(_ptr(buf), buf.size, as_int32(key))
    return buf


def cycle_at(buf, key, stream_off=0):
    """buf[j] ^= ks[stream_off + j] in place."""
    _oracle().oracle_cycle_at(_ptr(buf), buf.size, as_int32(key), stream_off)
    return buf


def state_at(key, i):
    return _oracle().oracle_state_at(as_int32(key), i)


def keystream_at(key, i):
    return _oracle().oracle_keystream_at(as_int32(key), i)


def keystream(key, n, stream_off=0):
    """ks[stream_off : stream_off+n] as a uint8 ndarray (Cycle over zero bytes)."""
    z = np.zeros(n, dtype=np.uint8)
    return cycle_at(z, key, stream_off)


def fnv1a64(buf, seed=FNV_OFFSET):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    return _oracle().oracle_fnv1a64(ctypes.c_void_p(buf.ctypes.data), buf.size, seed)


def hdr_decrypt(hdr):
    return _oracle().oracle_hdr_decrypt(_ptr(hdr), hdr.size)


def hdr_encrypt(hdr, ps4=True):
    return _oracle().oracle_hdr_encrypt(_ptr(hdr), hdr.size, 1 if ps4 else 0)


# --- the compiled reference itself (optional) ------------------------------------------
_ref = None


def have_ref():
    return os.path.exists(_REF_SO)


def ref_cycle(buf, key):
    """In-place Cycle by the reference's own object code (oracle/_ref)."""
    global _ref
    if _ref is None:
        lib = ctypes.CDLL(_REF_SO)
        lib.ref_cycle.restype = None
        lib.ref_cycle.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int]
        _ref = lib
    assert buf.size <= 0xFFFFFFFF
    _ref.ref_cycle(_ptr(buf), buf.size, as_int32(key))
    return buf


# --- pure-Python loops, small cases only (a third, independent statement) -------------
def pure_cycle_key(k):
    """CEncryptionCycler.cpp:16-25 with C truncating division on int32."""
    def tdiv(x, y):
        q = abs(x) // abs(y)
        return -q if (x < 0) != (y < 0) else q

    hi = tdiv(k, 127773)
    lo = k - hi * 127773
    t = lo * 16807 - hi * 2836
    assert -(1 << 31) <= t < (1 << 31)
    if t <= 0:
        t += 0x7FFFFFFF
    return t


def pure_cycle(data, key):
    k = pure_cycle_key(as_int32(key))
    out = bytearray(data)
    for i in range(len(out)):
        out[i] ^= (k ^ 0xFF) & 0xFF
        k = pure_cycle_key(k)
    return bytes(out)


def splitmix_bytes(n, seed):
    """SURVEY 8d synthetic plaintext: splitmix64 counter stream, 8 B per draw, little-endian."""
    k = (n + 7) // 8
    with np.errstate(over="ignore"):
        z = (np.arange(1, k + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.view(np.uint8)[:n].copy()
