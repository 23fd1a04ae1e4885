"""TEST INFRASTRUCTURE -- pure-Python restatement of the decrypted `.hdr` wire format and of the
part-split bookkeeping, used to check the C++ host mirror (modulate_amd/csrc/host/CArk.cpp).

PARITY UNPINNED: the reference has no tests or fixtures for this format and its CArk.cpp cannot
be built here (Win32).  Everything below follows the cited lines of
Modulate/CArk.cpp only:
    layout written          :911-1131        entry serialise   :685-721
    layout read             :341-416, 594-650
    bucket hash             :832-843         chain / bucket table  :1066-1131
    PS3 order               :1052-1063       PS4 order (path)  :977-1048 (ties by table position;
                                             the reference's comparator is not a strict weak order)
    part split              :783-823
The 16 `mChecksumData` bytes (header offsets 12..27) are uninitialised stack in the reference
(SURVEY F6); zeros here.
"""
import struct

MAGIC = {True: 0x6F303F55, False: 0xC64EED30}      # Settings.h:16-17 (ps4, ps3)
HASH_FIELD = {True: 0xDDB682F0, False: 0x7D401F60}  # CArk.cpp:719-720


def _tdiv(a, b):
    q = abs(a) // abs(b)
    return -q if (a < 0) != (b < 0) else q


def name_bucket(name, n_files):
    """CArk.cpp:832-843: int arithmetic on signed chars, do-while (an empty name hashes its NUL)."""
    h = 0
    data = name.encode("latin-1") or b"\0"
    for c in data:
        c = c - 256 if c >= 128 else c
        h = h * 0x7F + c
        h -= _tdiv(h, n_files) * n_files
    return h


def entry_order(names, flags1, flags2, ps4):
    n = len(names)
    buckets = [name_bucket(nm, n) for nm in names]
    if not ps4:
        return sorted(range(n), key=lambda i: (buckets[i], i)), buckets

    def key(i):
        comps = names[i].lower().split("/")
        # at each level: leaf (file) before non-leaf (directory), then the component name
        k = []
        for d, c in enumerate(comps):
            k.append((0 if d + 1 == len(comps) else 1, c))
        return (k, flags1[i], flags2[i], i)
    return sorted(range(n), key=key), buckets


def split_into_arks(sizes, planned):
    """CArk.cpp:783-823 from sizes alone -> (offsets, part_sizes)."""
    planned = list(planned)
    offsets = []
    idx, allowed, start, ptr = 0, planned[0], 0, 0
    for s in sizes:
        if s == 0:
            offsets.append(0)
            continue
        offsets.append(ptr)
        ptr += s
        if ptr - start > allowed and idx + 1 < len(planned):
            ark = ptr - start
            planned[idx] = ark
            idx += 1
            allowed += planned[idx] - ark
            start = ptr
    planned[idx] = ptr - start
    for j in range(idx + 1, len(planned)):
        planned[j] = 0
    return offsets, planned


def even_plan(total, n_arks):
    """CArk.cpp:207-217."""
    out, rem = [], total
    for i in range(n_arks):
        out.append(rem // (n_arks - i))
        rem -= out[-1]
    return out


def serialise(names, sizes, offsets, ark_sizes, ark_paths, ps4, flags1=None, flags2=None):
    """Decrypted header image (magic + plaintext body) as SaveArk lays it out."""
    n, na = len(names), len(ark_sizes)
    flags1 = flags1 or [-1] * n
    flags2 = flags2 or [-1] * n
    out = bytearray()
    out += struct.pack("<I", MAGIC[ps4])
    out += struct.pack("<II", 9, 1) + bytes(16) + struct.pack("<i", na)
    out += struct.pack("<i", na) + b"".join(struct.pack("<I", s) for s in ark_sizes)
    out += struct.pack("<i", na)
    for p in ark_paths:
        b = p.encode("latin-1")
        out += struct.pack("<i", len(b)) + b
    out += struct.pack("<i", na) + bytes(4 * na)
    out += struct.pack("<i", na) + bytes(4 * na)
    out += struct.pack("<i", n)
    order, buckets = entry_order(names, flags1, flags2, ps4)
    last = {}
    for idx, i in enumerate(order):
        link = last.get(buckets[i], -1)
        last[buckets[i]] = idx
        b = names[i].encode("latin-1")
        out += struct.pack("<q", offsets[i]) + struct.pack("<i", len(b)) + b
        out += struct.pack("<iII", link, sizes[i], HASH_FIELD[ps4] if sizes[i] else 0)
    out += struct.pack("<i", n)
    for h in range(n):
        out += struct.pack("<i", last.get(h, -1))
    return bytes(out)


def parse(image):
    """Decrypted header image -> dict, following CArk::Load's reading order (CArk.cpp:341-416)."""
    at = 0

    def u32():
        nonlocal at
        v = struct.unpack_from("<I", image, at)[0]
        at += 4
        return v

    def i32():
        nonlocal at
        v = struct.unpack_from("<i", image, at)[0]
        at += 4
        return v

    def string():
        nonlocal at
        ln = i32()
        s = image[at:at + min(ln, 255)].decode("latin-1")
        at += ln
        return s.split("\0")[0]

    magic = u32()
    version, n_checks = u32(), u32()
    at += 16
    na = i32()
    n_sizes = i32()
    ark_sizes = [u32() for _ in range(n_sizes)][:na]
    n_paths = i32()
    ark_paths = [string() for _ in range(n_paths)][:na]
    nc = i32()
    at += 4 * nc
    at += 4 * nc
    assert i32() == 0
    n = i32()
    files = []
    for _ in range(n):
        off = struct.unpack_from("<q", image, at)[0]
        at += 8
        name = string()
        f1, size, hsh = i32(), u32(), u32()
        files.append({"offset": off, "name": name, "flags1": f1, "size": size, "hash": hsh})
    n2 = i32()
    for i in range(n):
        files[i]["flags2"] = i32() if i < n2 else -1
    return {"magic": magic, "version": version, "num_checksums": n_checks, "ark_sizes": ark_sizes,
            "ark_paths": ark_paths, "files": files, "end": at}


def lookup(parsed, name):
    """Find `name` through the header's own bucket table + chain links (what the game does)."""
    files = parsed["files"]
    n = len(files)
    i = files[name_bucket(name, n)]["flags2"] if n else -1  # bucket head lives in the trailing list
    while i != -1:
        if files[i]["name"] == name:
            return i
        i = files[i]["flags1"]
    return -1
