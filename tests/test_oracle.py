"""CPU-only: pins the oracle (oracle/cycle_oracle.c) against

* the golden vectors emitted by the COMPILED REFERENCE (tests/golden/cycle_golden.json, made by
  oracle/make_golden.py over oracle/_ref), including the 2^32-1 byte run,
* the known-answer bytes recorded in SURVEY.md 8c,
* the compiled reference itself when oracle/_ref is present (this container), and
* a third, pure-Python statement of CEncryptionCycler.cpp:4-25 on small cases.
"""
import numpy as np
import pytest

# SURVEY.md 8c known-answer keystream prefixes (Cycle over zero bytes), captured from the
# compiled reference during the survey.
SURVEY_KATS = {
    0xC64EED30: "47c67c861d86ae508adee97f58b99d0bdbc024eabbf43a695a0569d75127b117",
    0x90CFC0AB: "7accad6faf91a7e372008f0719ba3403bc26c7122a8dd1592ae7a5b3f522b73a",
    0: "00" * 32, 0x7FFFFFFF: "00" * 32, 0x80000001: "00" * 32,
    1: "580e26d57d372701bcb267aa731d4cb8e8ee67abd0eed2faa70a942977f8666d",
    0xFFFFFFFF: "a7f1d92a82c8d8fe434d98558ce2b347171198542f112d0558f56bd688079992",
    0x80000000: "a7f1d92a82c8d8fe434d98558ce2b347171198542f112d0558f56bd688079992",
    0x7FFFFFFE: "a7f1d92a82c8d8fe434d98558ce2b347171198542f112d0558f56bd688079992",
    12345: "d0ff3f875ce8a3b553c8a906c10391a583d7aa12671894049a0abdd1a8a009bc",
    (-127772) & 0xFFFFFFFF: "4402dd8220281a9b5fef088a0ef942247eb3fd8c83a2e2eff4ca9ac881bc1563",
}


def test_survey_kats(oracle):
    for key, hexs in SURVEY_KATS.items():
        assert oracle.keystream(key, 32).tobytes().hex() == hexs, hex(key)


def test_survey_large_offsets_closed_form(oracle):
    P = oracle.PERIOD
    k = oracle.KEY_PS4
    assert oracle.keystream(k, 24, P - 8).tobytes().hex() == "7b25de2caf205155" + "7accad6faf91a7e372008f0719ba3403"
    assert oracle.keystream(k, 16, (1 << 32) - 17).tobytes().hex() == "4f73a901c67b25de2caf2051557accad"
    assert oracle.keystream_at(k, (1 << 32) - 1) == oracle.keystream_at(k, 3) == 0x6F


def test_golden_keystreams(oracle, golden):
    for e in golden["keystream"]:
        ks = oracle.keystream(e["key"], 1 << 20)
        assert ks[:64].tobytes().hex() == e["first64"]
        assert f"{oracle.fnv1a64(ks[:4096]):016x}" == e["fnv_4k"]
        assert f"{oracle.fnv1a64(ks):016x}" == e["fnv_1m"]
        assert ks[-16:].tobytes().hex() == e["at_1m_minus_16"]
    for e in golden["cycle_key"]:
        assert oracle.keystream(e["key"], 4).tobytes().hex() == e["ks4"]


def test_golden_plaintext_cases(oracle, golden):
    for e in golden["plaintext_cases"]:
        pt = oracle.splitmix_bytes(e["n"], e["seed"])
        assert f"{oracle.fnv1a64(pt):016x}" == e["pt_fnv"]
        ct = oracle.cycle(pt.copy(), e["key"])
        assert f"{oracle.fnv1a64(ct):016x}" == e["ct_fnv"], e
        assert ct[:16].tobytes().hex() == e["ct_first16"] and ct[-16:].tobytes().hex() == e["ct_last16"]
        assert np.array_equal(oracle.cycle(ct, e["key"]), pt)  # involution
    b = ((np.arange(4096, dtype=np.uint32) * 131 + 7) & 0xFF).astype(np.uint8)
    assert f"{oracle.fnv1a64(oracle.cycle(b.copy(), golden['survey_4k']['key'])):016x}" == golden["survey_4k"]["ct_fnv"]


def test_golden_large_samples_closed_form(oracle, golden):
    """The jump-ahead (closed form) agrees with the reference's 2^32-1 byte serial run."""
    L = golden["large"]
    assert L is not None and L["period_repeats_1g"] is True
    k = L["key"]
    for s in L["samples"] + [L["around_period"] | {"off": L["around_period"]["start"]},
                             L["tail16"] | {"off": L["tail16"]["start"]}]:
        n = len(s["hex"]) // 2
        assert oracle.keystream(k, n, s["off"]).tobytes().hex() == s["hex"], s["off"]
    assert f"{oracle.fnv1a64(oracle.keystream(k, 1 << 20, 1 << 31)):016x}" == L["fnv_at_2g_1m"]
    assert f"{oracle.fnv1a64(oracle.keystream(k, 4096)):016x}" == L["fnv_4k"]


def test_state_at_matches_serial(oracle):
    rng = np.random.default_rng(1)
    for key in [oracle.KEY_PS3, oracle.KEY_PS4, 1, 2, 0xFFFFFFFF, 0x80000000, 12345, 0, 0x7FFFFFFF]:
        k = oracle.cycle_key(key)
        for i in range(300):
            assert oracle.state_at(key, i) == k, (hex(key), i)
            k = oracle.cycle_key(k)
        for off in rng.integers(0, 1 << 40, size=8):
            off = int(off)
            a = oracle.keystream(key, 100, off)
            b = np.array([oracle.keystream_at(key, off + j) for j in range(100)], dtype=np.uint8)
            assert np.array_equal(a, b)


def test_pure_python_statement(oracle):
    for key in [oracle.KEY_PS3, oracle.KEY_PS4, 0, 1, 0xFFFFFFFF, 0x80000000, 0x80000001, (-127772) & 0xFFFFFFFF]:
        data = bytes(range(256)) * 3
        assert oracle.pure_cycle(data, key) == oracle.cycle(np.frombuffer(data, dtype=np.uint8).copy(), key).tobytes()
    for k in [-(1 << 31), -1, 0, 1, 127772, 127773, 127774, 0x7FFFFFFF, -127773, 0x7FFFFFFE]:
        assert oracle.pure_cycle_key(k) == oracle.cycle_key(k), k


def test_cycle_key_range_and_zero_residue(oracle):
    # keys == 0 mod m stick at m, keystream all zero
    for key in (0, 0x7FFFFFFF, 0x80000001):
        assert oracle.cycle_key(key) == 0x7FFFFFFF
        assert not oracle.keystream(key, 1000).any()
    rng = np.random.default_rng(7)
    for k in rng.integers(-(1 << 31), 1 << 31, size=2000):
        r = oracle.cycle_key(int(k))
        assert 1 <= r <= 0x7FFFFFFF and r % 0x7FFFFFFF == (16807 * int(k)) % 0x7FFFFFFF


def test_serial64_and_window(oracle):
    pt = oracle.splitmix_bytes(100000, 3)
    a = oracle.cycle(pt.copy(), oracle.KEY_PS4)
    assert np.array_equal(oracle.cycle_serial64(pt.copy(), oracle.KEY_PS4), a)
    w = pt[5000:9000].copy()
    assert np.array_equal(oracle.cycle_at(w, oracle.KEY_PS4, 5000), a[5000:9000])


def test_header_framing(oracle):
    body = oracle.splitmix_bytes(4092, 11)
    for ps4, magic, key in ((True, oracle.MAGIC_PS4, oracle.KEY_PS4), (False, oracle.MAGIC_PS3, oracle.KEY_PS3)):
        hdr = np.concatenate([np.zeros(4, np.uint8), body])
        assert oracle.hdr_encrypt(hdr, ps4) == 0
        assert int.from_bytes(hdr[:4].tobytes(), "little") == magic
        assert np.array_equal(hdr[4:], oracle.cycle(body.copy(), key))
        assert oracle.hdr_decrypt(hdr) == 0
        assert np.array_equal(hdr[4:], body)
    bad = np.zeros(64, np.uint8)
    assert oracle.hdr_decrypt(bad) == 3  # eError_UnknownVersionNumber


def test_against_compiled_reference(oracle):
    """Direct comparison with the reference's object code (only where oracle/_ref exists)."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no reference sources on this machine)")
    rng = np.random.default_rng(5)
    keys = [oracle.KEY_PS3, oracle.KEY_PS4, 0, 1, 0x7FFFFFFF, 0x80000000, 0x80000001, 0xFFFFFFFF] + \
        [int(x) for x in rng.integers(0, 1 << 32, size=24)]
    for key in keys:
        n = int(rng.integers(0, 70000))
        pt = rng.integers(0, 256, size=n, dtype=np.uint8)
        assert np.array_equal(oracle.ref_cycle(pt.copy(), key), oracle.cycle(pt.copy(), key)), hex(key)
    pt = oracle.splitmix_bytes(1 << 24, 99)
    assert np.array_equal(oracle.ref_cycle(pt.copy(), oracle.KEY_PS4), oracle.cycle(pt.copy(), oracle.KEY_PS4))
