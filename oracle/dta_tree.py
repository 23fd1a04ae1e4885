"""TEST INFRASTRUCTURE -- pure-Python restatement of the binary DTA tree format, used to check
modulate_amd/csrc/host/CDtaFile.cpp.  PARITY UNPINNED (no reference tests; CDtaFile.cpp needs Win32
headers through Utils.h).  Follows Modulate/CDtaFile.cpp:57-100, 393-509 (load) and
:362-391, 1302-1326 + CDtaFile.h:262-284 (save).

A node is a tuple:  ("tree", type(16|17), node_id, [children])  |  ("int", type, value)
                    ("float", 1, f32-bits)                      |  ("str", type, text)
"""
import struct

INT_TYPES, STR_TYPES, TREE_TYPES = (0, 6, 8, 9), (5, 18, 33, 35), (16, 17)


def _write_body(node, out):
    _, _, node_id, children = node
    out += struct.pack("<hh", len(children), node_id)
    for c in children:
        out += struct.pack("<i", c[1])
        if c[0] == "tree":
            out += struct.pack("<i", 1)
            _write_body(c, out)
        elif c[0] == "float":
            out += struct.pack("<I", c[2])
        elif c[0] == "str":
            b = c[2].encode("latin-1")
            out += struct.pack("<i", len(b)) + b
        else:
            out += struct.pack("<i", c[2])


def serialise(top_level, separators=True):
    """separators=True: the form the reference's Load reads (type, 1 between top-level trees,
    CDtaFile.cpp:95-96).  separators=False: what the reference's Save actually writes -- top-level
    trees back to back (CDtaFile.cpp:371-374)."""
    out = bytearray(b"\x01" + struct.pack("<i", 1))
    for k, n in enumerate(top_level):
        if k and separators:
            out += struct.pack("<ii", n[1], 1)
        _write_body(n, out)
    return bytes(out)


def _read_body(data, at, ttype):
    n, node_id = struct.unpack_from("<hh", data, at)
    at += 4
    if n <= 0:
        raise ValueError("tree with no children")
    children = []
    for _ in range(n):
        (t,) = struct.unpack_from("<i", data, at)
        at += 4
        if t in STR_TYPES:
            (ln,) = struct.unpack_from("<i", data, at)
            at += 4
            if ln < 0 or at + ln > len(data):
                raise ValueError("bad string")
            children.append(("str", t, data[at:at + ln].decode("latin-1").split("\0")[0]))
            at += ln
        elif t in TREE_TYPES:
            at += 4
            sub, at = _read_body(data, at, t)
            children.append(sub)
        elif t in INT_TYPES:
            children.append(("int", t, struct.unpack_from("<i", data, at)[0]))
            at += 4
        elif t == 1:
            children.append(("float", 1, struct.unpack_from("<I", data, at)[0]))
            at += 4
        else:
            raise ValueError(f"bad node type {t}")
    return ("tree", ttype, node_id, children), at


def parse(data):
    at, ttype, top = 5, 16, []
    while at < len(data):
        if ttype not in TREE_TYPES:
            raise ValueError("bad top-level type")
        node, at = _read_body(data, at, ttype)
        top.append(node)
        if at >= len(data):
            break
        ttype, _ = struct.unpack_from("<ii", data, at)
        at += 8
    return top


def dump(top_level):
    """Same text as CDtaFile::Dump()."""
    lines = []

    def rec(n, d):
        pad = "  " * d
        if n[0] == "tree":
            lines.append(f"{pad}{n[1]} tree id={n[2]} n={len(n[3])}")
            for c in n[3]:
                rec(c, d + 1)
        elif n[0] == "float":
            lines.append(f"{pad}1 f32bits={n[2]}")
        elif n[0] == "str":
            lines.append(f"{pad}{n[1]} str={n[2]}")
        else:
            lines.append(f"{pad}{n[1]} int={n[2]}")
    for n in top_level:
        rec(n, 0)
    return "\n".join(lines) + ("\n" if lines else "")


def synth_tree(rng, target_bytes=4000):
    """A DTA-shaped blob like amp_config: one top-level tree of per-song sub-trees."""
    kids = []
    i = 0
    while True:
        kids.append(("tree", 16, i + 2, [("str", 18, f"song_{i}"), ("str", 5, f"songs/s{i}/s{i}.moggsong"),
                                         ("float", 1, int(rng.integers(0x3F800000, 0x43000000))),
                                         ("int", 0, int(rng.integers(-1000, 1000))), ("int", 6, i), ("int", 8, 1), ("int", 9, 0),
                                         ("tree", 17, i + 2, [("str", 35, "DEF"), ("str", 33, "inc.dta"), ("str", 5, "")])]))
        i += 1
        if len(serialise([("tree", 16, 1, kids)])) >= target_bytes or i >= 500:
            return [("tree", 16, 1, kids)]
