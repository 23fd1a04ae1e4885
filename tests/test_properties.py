"""Property-based checks (hypothesis), CPU only: the product's host loop against the oracle over random keys,
lengths, alignments and 64-bit stream offsets; the algebra the GPU path relies on (involution, a stream cut at any
byte equals one call, periodicity); the C++ header writer against the Python restatement on random tables; the C++ DTA
tree reader and writer against theirs on random trees; pack -> save -> load -> extract on random tables."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import ark_header as AH

needs_host_loop = pytest.mark.skipif(os.environ.get("MODGPU_REQUIRE_GPU", "0") not in ("", "0"),
                                     reason="MODGPU_REQUIRE_GPU forbids the host loop in this process")
COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
keys = st.one_of(st.sampled_from([0x90CFC0AB, 0xC64EED30, 0, 1, 0x7FFFFFFF, 0x80000000, 0x80000001, 0xFFFFFFFF, 0x7FFFFFFE]),
                 st.integers(0, 0xFFFFFFFF))
offsets = st.one_of(st.integers(0, 1 << 20), st.integers(0, (1 << 64) - 1),
                    st.sampled_from([0x7FFFFFFE - 3, 0x7FFFFFFE, (1 << 32) - 1, 1 << 32, (1 << 64) - 1]))


@needs_host_loop
@settings(max_examples=150, **COMMON)
@given(key=keys, n=st.integers(0, 3000), off=offsets, lead=st.integers(0, 17), seed=st.integers(0, 1 << 30))
def test_host_loop_equals_oracle(modgpu, oracle, key, n, off, lead, seed):
    whole = oracle.splitmix_bytes(n + lead + 9, seed)
    got, want = whole.copy(), whole.copy()
    modgpu.cycle_scalar_host(got[lead:lead + n], key, off)
    oracle.cycle_at(want[lead:lead + n], key, off)
    assert np.array_equal(got, want)


@needs_host_loop
@settings(max_examples=150, **COMMON)
@given(isa=st.sampled_from(["generic", "avx2", "avx512"]), key=keys, n=st.integers(0, 700), off=offsets,
       lead=st.integers(0, 70), seed=st.integers(0, 1 << 30))
def test_each_host_loop_body_equals_oracle(modgpu, oracle, isa, key, n, off, lead, seed):
    """The three bodies of scalar_path.cpp by name, at any alignment against their 16/32/64-byte blocks."""
    whole = oracle.splitmix_bytes(n + lead + 9, seed)
    got, want = whole.copy(), whole.copy()
    try:
        modgpu.cycle_scalar_host_isa(got[lead:lead + n], key, isa, off)
    except modgpu.ModGpuError as e:
        assert e.code == 1  # this CPU does not run that body
        return
    oracle.cycle_at(want[lead:lead + n], key, off)
    assert np.array_equal(got, want)


@needs_host_loop
@settings(max_examples=60, **COMMON)
@given(key=keys, n=st.integers(1, 5000), off=offsets, cut=st.floats(0, 1), seed=st.integers(0, 1 << 30))
def test_involution_and_split_stream(modgpu, oracle, key, n, off, cut, seed):
    pt = oracle.splitmix_bytes(n, seed)
    one = modgpu.cycle_scalar_host(pt.copy(), key, off)
    assert np.array_equal(modgpu.cycle_scalar_host(one.copy(), key, off), pt)  # the cipher is its own inverse
    k = int(cut * n)
    two = pt.copy()
    modgpu.cycle_scalar_host(two[:k], key, off)
    modgpu.cycle_scalar_host(two[k:], key, (off + k) & ((1 << 64) - 1))  # a stream cut at any byte == one call
    if off + k < (1 << 64):  # (positions do not wrap at 2^64 in the reference's terms: skip the wrap case)
        assert np.array_equal(two, one)


@settings(max_examples=200, **COMMON)
@given(key=keys, i=st.integers(0, (1 << 64) - 1))
def test_state_at_matches_oracle_and_period(modgpu, oracle, key, i):
    s = modgpu.state_at(key, i)
    assert s == oracle.state_at(key, i) & 0xFFFFFFFF
    assert 1 <= s <= 0x7FFFFFFF
    if i + oracle.PERIOD < (1 << 64):
        assert modgpu.state_at(key, i + oracle.PERIOD) == s  # the keystream has period 2^31 - 2


names = st.lists(st.text(alphabet="abcXYZ019_./", min_size=1, max_size=40).filter(
    lambda s: not s.startswith("/") and not s.endswith("/") and "//" not in s), min_size=1, max_size=60, unique=True)


@settings(max_examples=80, **COMMON)
@given(names=names, data=st.data())
def test_header_writer_equals_restatement_on_random_tables(names, data):
    from modulate_amd import host as H
    H.lib()
    H.set_flags(overwrite=True, ignore_new=True, pack_all=False, verbose=False)
    sizes = [data.draw(st.integers(0, 5000)) for _ in names]
    n_arks = data.draw(st.integers(1, 5))
    ps4 = data.draw(st.booleans())
    H.select_platform(ps4)
    try:
        a = H.Ark()
        a.construct_from_table(names, sizes, n_arks, "p")
        a.build_from_memory(np.zeros(sum(sizes), np.uint8))
        offs, parts = AH.split_into_arks(sizes, AH.even_plan(sum(sizes), n_arks))
        assert a.ark_sizes() == parts and [f["offset"] for f in a.files()] == offs
        img = a.serialise_header(encrypt=False).tobytes()
        assert img == AH.serialise(names, sizes, offs, parts, a.ark_paths(), ps4)
        p = AH.parse(img)
        assert p["end"] == len(img) and sorted(f["name"] for f in p["files"]) == sorted(names)
        for nm in names:
            assert p["files"][AH.lookup(p, nm)]["name"] == nm  # the header's own hash chains find every entry
        if os.environ.get("MODGPU_REQUIRE_GPU", "0") in ("", "0"):  # encrypt + decrypt through Cycle (host loop here), then parse
            b = H.Ark()
            b.parse_header(a.serialise_header(encrypt=True))
            assert [(f["name"], f["size"], f["offset"], f["flags1"]) for f in b.files()] == \
                [(f["name"], f["size"], f["offset"], f["flags1"]) for f in p["files"]]
            assert b.ark_sizes() == parts
            b.close()
        a.close()
    finally:
        H.select_platform(True)


# ---- binary DTA tree: C++ reader + writer against the restatement on random trees ----------------------
_leaf = st.one_of(
    st.tuples(st.just("int"), st.sampled_from([0, 6, 8, 9]), st.integers(-(1 << 31), (1 << 31) - 1)),
    st.tuples(st.just("float"), st.just(1), st.integers(0, (1 << 32) - 1)),
    st.tuples(st.just("str"), st.sampled_from([5, 18, 33, 35]),
              st.text(alphabet=st.characters(min_codepoint=1, max_codepoint=255), max_size=40)))  # latin-1, no NUL (C strings upstream)


def _tree(children):
    return st.tuples(st.just("tree"), st.sampled_from([16, 17]), st.integers(-(1 << 15), (1 << 15) - 1),
                     st.lists(children, min_size=1, max_size=6))


_node = st.recursive(_leaf, _tree, max_leaves=40)


@settings(max_examples=120, **COMMON)
@given(top=st.lists(_tree(_node), min_size=1, max_size=4))
def test_dta_reader_and_writer_equal_restatement_on_random_trees(top):
    """Typed node trees of any shape (CDtaFile.cpp:57-100, 393-509 load; 362-391, 1302-1326 save): the C++ parse of the
    restatement's image dumps node for node like the restatement's own tree, and the C++ writer reproduces the image
    byte for byte -- several top-level trees in the form Load reads (separators), which Save writes only with the quirk fix."""
    from modulate_amd import host as H
    from oracle import dta_tree as DT
    top = [(t[0], 16 if k == 0 else t[1], t[2], t[3]) for k, t in enumerate(top)]  # the first top-level tree is type 16 by construction (CDtaFile.cpp:57-60)
    blob = DT.serialise(top)
    assert DT.parse(blob) == top
    H.lib()
    H.set_fix_quirks(True)
    try:
        out, dump = H.dta_roundtrip(blob)
    finally:
        H.set_fix_quirks(False)
    assert dump == DT.dump(top)
    assert out == blob
    if len(top) == 1:  # one top-level tree: the reference's own Save form is the same image
        out2, _ = H.dta_roundtrip(blob)
        assert out2 == blob


# ---- pack -> save -> load -> extract on random tables, with and without the part cipher ---------------
# (directories in upper case, file names in lower case: no name is both a file and a directory of another entry)
fs_names = st.lists(st.tuples(st.lists(st.text(alphabet="ABC", min_size=1, max_size=3), max_size=3), st.text(alphabet="abc019_", min_size=1, max_size=12))
                    .map(lambda t: "/".join(t[0] + [t[1]])), min_size=1, max_size=40, unique=True)


@needs_host_loop
@settings(max_examples=40, **COMMON)
@given(names=fs_names, data=st.data())
def test_pack_save_load_extract_on_random_tables(names, data):
    """Entry table -> part buffers -> header + part files on disk -> Load -> LoadArkData -> ExtractFiles: every file comes back
    byte for byte, whatever the split into parts, on both platforms, with the parts stored plain (the reference) or cycled."""
    import tempfile
    from modulate_amd import host as H
    H.lib()
    H.set_flags(overwrite=True, ignore_new=True, pack_all=False, verbose=False)
    sizes = [data.draw(st.integers(0, 3000)) for _ in names]
    n_arks = data.draw(st.integers(1, 4))
    ps4 = data.draw(st.booleans())
    from modulate_amd import capi
    crypt = data.draw(st.booleans()) and capi.device_count() > 0  # (the part cipher is the GPU's: it fails loudly without one)
    seed = data.draw(st.integers(0, 1 << 30))
    payload = np.random.default_rng(seed).integers(0, 256, size=sum(sizes), dtype=np.uint8)
    plat = "ps4" if ps4 else "ps3"
    H.select_platform(ps4)
    H.set_fix_quirks(True)  # (SaveArk without a placeholder header in the working directory; ExtractFiles honours its range)
    try:
        with tempfile.TemporaryDirectory() as d:
            a = H.Ark()
            a.construct_from_table(names, sizes, n_arks, f"main_{plat}")
            a.build_from_memory(payload)
            a.enable_part_cipher(crypt)
            os.makedirs(d + "/packed")
            a.save(d + "/packed/", f"main_{plat}.hdr")
            table = {f["name"]: (f["offset"], f["size"]) for f in a.files()}
            a.close()
            b = H.Ark().load(d + f"/packed/main_{plat}.hdr")
            b.enable_part_cipher(crypt)
            b.load_data()
            b.extract(d + "/out/")
            b.close()
            for nm, (off, size) in table.items():
                got = np.fromfile(d + "/out/" + nm, dtype=np.uint8) if size else np.zeros(0, np.uint8)
                assert np.array_equal(got, payload[off:off + size]), nm
    finally:
        H.set_fix_quirks(False)
        H.select_platform(True)
