"""ctypes binding over libmodulate_host.so (the C++ host mirror: CEncryptionCycler, CArk, the
Decode/Unpack/Pack commands) through its test/bench hooks (modulate_amd/csrc/host/host_capi.h)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u64, _u32, _i32, _i64, _int, _vp, _cp = (ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p)

ERRORS = ["NoError", "FailedToOpenFile", "FailedToCreateDirectory", "UnknownVersionNumber", "ValueOutOfBounds",
          "AlreadyLoaded", "InvalidData", "NoData", "FailedToCreateFile", "FailedToDeleteFile", "FailedToCopyFile",
          "InvalidParameter", "FailedToWriteData"]  # Modulate/Error.h:5-20

HOOKS = {
    "modhost_last_error": (_cp, []),
    "modhost_select_platform": (None, [_int]),
    "modhost_set_flags": (None, [_int, _int, _int, _int]),
    "modhost_set_fix_quirks": (None, [_int]),
    "modhost_cycle_via_class": (_int, [_vp, _u32, _i32, _int]),
    "modhost_decode": (_int, [_cp]),
    "modhost_dta_roundtrip": (_int, [_vp, _u64, _vp, _u64, ctypes.POINTER(_u64), _cp, _u64]),
    "modhost_ark_new": (_vp, []),
    "modhost_ark_free": (None, [_vp]),
    "modhost_ark_load": (_int, [_vp, _cp]),
    "modhost_ark_parse_header": (_int, [_vp, _vp, _u64]),
    "modhost_ark_load_data": (_int, [_vp]),
    "modhost_ark_extract": (_int, [_vp, _int, _int, _cp]),
    "modhost_ark_construct_from_directory": (_int, [_vp, _cp, _vp]),
    "modhost_ark_construct_from_table": (_int, [_vp, _cp, ctypes.POINTER(_u32), _int, _int, _cp]),
    "modhost_ark_build": (_int, [_vp, _cp]),
    "modhost_ark_build_from_memory": (_int, [_vp, _vp, _u64]),
    "modhost_ark_save": (_int, [_vp, _cp, _cp]),
    "modhost_ark_cycle_parts": (_int, [_vp, _i32, _int]),
    "modhost_ark_enable_part_cipher": (None, [_vp, _int, _int]),
    "modhost_ark_serialise_header": (_int, [_vp, _int, _vp, _u64, ctypes.POINTER(_u64)]),
    "modhost_ark_num_files": (_int, [_vp]),
    "modhost_ark_num_arks": (_int, [_vp]),
    "modhost_ark_ark_size": (_u32, [_vp, _int]),
    "modhost_ark_ark_path": (_cp, [_vp, _int]),
    "modhost_ark_file_name": (_cp, [_vp, _int]),
    "modhost_ark_file_size": (_u32, [_vp, _int]),
    "modhost_ark_file_offset": (_i64, [_vp, _int]),
    "modhost_ark_file_flags1": (_int, [_vp, _int]),
    "modhost_ark_file_flags2": (_int, [_vp, _int]),
    "modhost_ark_data_size": (_u64, [_vp]),
    "modhost_ark_data": (_vp, [_vp]),
    "modhost_ark_data_pinned": (_int, [_vp]),
}


class HostError(RuntimeError):
    def __init__(self, code, text=""):
        name = ERRORS[code] if 0 <= code < len(ERRORS) else "exception"
        super().__init__(f"eError {code} ({name}) {text}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        # MODULATE_HOST_LIB: alternative build of the same library (the ASan/UBSan one, tests/test_sanitizers.py)
        path = os.environ.get("MODULATE_HOST_LIB") or os.path.join(_HERE, "libmodulate_host.so")
        if not os.path.exists(path):
            raise HostError(-1, f"{path} not built: make -C modulate_amd/csrc")
        L = ctypes.CDLL(path)
        for name, (res, args) in HOOKS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise HostError(rc, lib().modhost_last_error().decode() if rc == -1 else "")


def select_platform(ps4=True):
    lib().modhost_select_platform(1 if ps4 else 0)


def set_flags(overwrite=True, ignore_new=True, pack_all=False, verbose=False):
    lib().modhost_set_flags(int(overwrite), int(ignore_new), int(pack_all), int(verbose))


def set_fix_quirks(on):
    """CSettings::mbFixReferenceQuirks: False (default) = the reference's behaviour, quirks included."""
    lib().modhost_set_fix_quirks(int(bool(on)))


def cycle_via_class(buf, key, device=-1):
    """CEncryptionCycler().Cycle(buf, n, key) -- the reference's own seam."""
    assert buf.dtype == np.uint8 and buf.flags["C_CONTIGUOUS"]
    key &= 0xFFFFFFFF
    key = key - (1 << 32) if key & 0x80000000 else key
    _check(lib().modhost_cycle_via_class(_vp(buf.ctypes.data), buf.size, key, device))
    return buf


def decode(directory):
    _check(lib().modhost_decode(os.fsencode(directory)))


def dta_roundtrip(blob):
    """CDtaFile: parse a binary DTA image, return (re-serialised bytes, text dump)."""
    blob = np.ascontiguousarray(np.frombuffer(bytes(blob), dtype=np.uint8))
    out = np.empty(max(1, 2 * blob.size + 64), dtype=np.uint8)
    size = _u64(0)
    dump = ctypes.create_string_buffer(max(1 << 16, 64 * blob.size))
    _check(lib().modhost_dta_roundtrip(_vp(blob.ctypes.data), blob.size, _vp(out.ctypes.data), out.size,
                                       ctypes.byref(size), dump, len(dump)))
    return out[:size.value].tobytes(), dump.value.decode("latin-1")


class Ark:
    """One CArk object."""

    def __init__(self):
        self.h = _vp(lib().modhost_ark_new())

    def close(self):
        if self.h:
            lib().modhost_ark_free(self.h)
            self.h = None

    __del__ = close

    def load(self, header_path):
        _check(lib().modhost_ark_load(self.h, os.fsencode(header_path)))
        return self

    def parse_header(self, image):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        _check(lib().modhost_ark_parse_header(self.h, _vp(image.ctypes.data), image.size))
        return self

    def load_data(self):
        _check(lib().modhost_ark_load_data(self.h))

    def extract(self, target_dir, first=0, count=None):
        _check(lib().modhost_ark_extract(self.h, first, self.num_files if count is None else count,
                                         os.fsencode(target_dir)))

    def construct_from_directory(self, input_dir, reference):
        _check(lib().modhost_ark_construct_from_directory(self.h, os.fsencode(input_dir), reference.h))

    def construct_from_table(self, names, sizes, n_arks, prefix="main"):
        if len(names) != len(sizes):
            raise HostError(11, "names and sizes differ in length")
        blob = b"".join(n.encode("latin-1") + b"\0" for n in names)
        arr = (_u32 * len(sizes))(*sizes)
        _check(lib().modhost_ark_construct_from_table(self.h, blob, arr, len(names), n_arks, prefix.encode()))

    def build(self, input_dir):
        _check(lib().modhost_ark_build(self.h, os.fsencode(input_dir)))

    def build_from_memory(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        _check(lib().modhost_ark_build_from_memory(self.h, _vp(data.ctypes.data), data.size))

    def save(self, output_dir, header_name):
        _check(lib().modhost_ark_save(self.h, os.fsencode(output_dir), header_name.encode()))

    def cycle_parts(self, key, n_devices=0):
        key &= 0xFFFFFFFF
        _check(lib().modhost_ark_cycle_parts(self.h, key - (1 << 32) if key & 0x80000000 else key, n_devices))

    def enable_part_cipher(self, on=True, n_devices=0):
        lib().modhost_ark_enable_part_cipher(self.h, int(on), n_devices)

    def serialise_header(self, encrypt):
        size = _u64(0)
        _check(lib().modhost_ark_serialise_header(self.h, int(encrypt), None, 0, ctypes.byref(size)))
        out = np.empty(size.value, dtype=np.uint8)
        _check(lib().modhost_ark_serialise_header(self.h, int(encrypt), _vp(out.ctypes.data), out.size, ctypes.byref(size)))
        return out

    num_files = property(lambda s: lib().modhost_ark_num_files(s.h))
    num_arks = property(lambda s: lib().modhost_ark_num_arks(s.h))

    def ark_sizes(self):
        return [lib().modhost_ark_ark_size(self.h, i) for i in range(self.num_arks)]

    def ark_paths(self):
        return [lib().modhost_ark_ark_path(self.h, i).decode("latin-1") for i in range(self.num_arks)]

    def files(self):
        L = lib()
        return [{"name": L.modhost_ark_file_name(self.h, i).decode("latin-1"), "size": L.modhost_ark_file_size(self.h, i),
                 "offset": L.modhost_ark_file_offset(self.h, i), "flags1": L.modhost_ark_file_flags1(self.h, i),
                 "flags2": L.modhost_ark_file_flags2(self.h, i)} for i in range(self.num_files)]

    @property
    def data_pinned(self):
        return bool(lib().modhost_ark_data_pinned(self.h))

    def data(self):
        n = lib().modhost_ark_data_size(self.h)
        if n == 0:
            return np.empty(0, np.uint8)
        p = lib().modhost_ark_data(self.h)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)).copy()
