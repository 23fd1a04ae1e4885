"""CPU-only checks of the C++ host mirror (libmodulate_host.so): it loads, the CArk header writer
and part-split bookkeeping agree with the independent Python restatement (oracle/ark_header.py;
parity unpinned -- the reference has no fixtures for this format), and anything that needs the
cipher fails loudly without a GPU instead of computing on the host."""
import os

import numpy as np
import pytest

from oracle import ark_header as AH


@pytest.fixture(scope="module")
def host():
    from modulate_amd import host as H
    H.lib()
    H.set_flags(overwrite=True, ignore_new=True, pack_all=False, verbose=False)
    return H


def synth_table(n, seed=1):
    rng = np.random.default_rng(seed)
    names = [f"dir{k % 97}/sub{k % 13}/f{k}.bin" for k in range(n)]  # SURVEY 8d config-4 naming
    names[: min(n, 3)] = ["readme.txt", "Dir5/UPPER.bin", "dir5/lower.bin"][: min(n, 3)]
    sizes = [int(x) for x in rng.integers(0, 3000, size=n)]
    if n > 10:
        sizes[4] = sizes[9] = 0  # empty files keep offset 0 (CArk.cpp:787-791)
    return names, sizes


def test_hooks_exported(host):
    L = host.lib()
    for name in host.HOOKS:
        assert getattr(L, name) is not None


@pytest.mark.parametrize("ps4", [True, False])
@pytest.mark.parametrize("n,n_arks", [(1, 1), (2, 1), (57, 3), (1000, 4), (5000, 8)])
def test_header_writer_matches_restatement(host, ps4, n, n_arks):
    host.select_platform(ps4)
    names, sizes = synth_table(n, seed=n)
    a = host.Ark()
    a.construct_from_table(names, sizes, n_arks, "main_ps4" if ps4 else "main_ps3")
    data = np.random.default_rng(n).integers(0, 256, size=sum(sizes), dtype=np.uint8)
    a.build_from_memory(data)
    offs, parts = AH.split_into_arks(sizes, AH.even_plan(sum(sizes), n_arks))
    assert a.ark_sizes() == parts and sum(parts) == sum(sizes)
    assert [f["offset"] for f in a.files()] == offs
    assert np.array_equal(a.data(), data)
    img = a.serialise_header(encrypt=False).tobytes()
    assert img == AH.serialise(names, sizes, offs, parts, a.ark_paths(), ps4)
    p = AH.parse(img)
    assert p["end"] == len(img) and p["magic"] == AH.MAGIC[ps4] and p["version"] == 9
    assert p["ark_sizes"] == parts and [f["name"] for f in p["files"]] != []
    # the header's own lookup structure finds every name, and only those
    for nm in names:
        i = AH.lookup(p, nm)
        assert i >= 0 and p["files"][i]["name"] == nm
    assert AH.lookup(p, "no/such/file") == -1
    by_name = {f["name"]: f for f in p["files"]}
    for nm, s, o in zip(names, sizes, offs):
        assert by_name[nm]["size"] == s and by_name[nm]["offset"] == o
        assert by_name[nm]["hash"] == (AH.HASH_FIELD[ps4] if s else 0)
    a.close()


def test_files_never_straddle_parts(host):
    host.select_platform(True)
    names, sizes = synth_table(3000, seed=5)
    a = host.Ark()
    a.construct_from_table(names, sizes, 6, "p")
    a.build_from_memory(np.zeros(sum(sizes), np.uint8))
    bounds = np.cumsum([0] + a.ark_sizes())
    for f in a.files():
        if f["size"]:
            k = np.searchsorted(bounds, f["offset"], side="right") - 1
            assert f["offset"] + f["size"] <= bounds[k + 1]
    a.close()


def test_bad_arguments(host):
    a = host.Ark()
    with pytest.raises(host.HostError) as e:
        a.construct_from_table(["a", "b"], [1], 1)
    assert e.value.code == 11  # eError_InvalidParameter
    a.construct_from_table(["a", "b"], [1, 2], 1)
    with pytest.raises(host.HostError) as e:
        a.build_from_memory(np.zeros(2, np.uint8))  # 3 bytes expected
    assert e.value.code == 11
    with pytest.raises(host.HostError) as e:
        host.Ark().load("/nonexistent/main_ps4.hdr")
    assert e.value.code == 1  # eError_FailedToOpenFile
    with pytest.raises(host.HostError) as e:
        host.Ark().parse_header(np.zeros(64, np.uint8))
    assert e.value.code == 3  # eError_UnknownVersionNumber: magic is checked before any cipher work
    a.close()


def test_cipher_needs_gpu(host, modgpu):
    """No CPU fallback behind the C++ seam: Cycle throws when the HIP path is unavailable."""
    if modgpu.device_count() > 0:
        pytest.skip("GPU present")
    buf = np.arange(100, dtype=np.uint8)
    keep = buf.copy()
    with pytest.raises(host.HostError) as e:
        host.cycle_via_class(buf, 0x90CFC0AB)
    assert e.value.code == -1 and "GPU path failed" in str(e.value) and np.array_equal(buf, keep)
    a = host.Ark()
    a.construct_from_table(["x"], [4], 1)
    a.build_from_memory(np.zeros(4, np.uint8))
    with pytest.raises(host.HostError):
        a.serialise_header(encrypt=True)
    a.close()


def test_cli_usage_and_unknown_flag():
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "modulate_amd", "bin", "modulate")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "usage" in r.stdout
    r = subprocess.run([exe, "-bogus"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unkown parameter" in r.stdout  # the reference's own spelling (Modulate.cpp:962)
    r = subprocess.run([exe, "-decode", "/nonexistent"], capture_output=True, text=True)
    assert r.returncode != 0 and "ERROR: Failed to open file" in r.stdout


# ---- binary DTA tree (parity unpinned) -------------------------------------
def test_dta_roundtrip_matches_restatement(host):
    from oracle import dta_tree as DT
    rng = np.random.default_rng(12)
    tree = DT.synth_tree(rng, target_bytes=4000)
    blob = DT.serialise(tree)
    assert DT.parse(blob) == tree
    out, dump = host.dta_roundtrip(blob)
    assert out == blob                      # C++ writer == Python writer on the C++ parse of the Python image
    assert dump == DT.dump(tree)            # node for node
    # several top-level trees (the reference's Save cannot round-trip these; ours writes the separators Load expects)
    multi = [tree[0], ("tree", 17, 9, [("int", 0, 5)]), ("tree", 16, 3, [("str", 5, "x"), ("float", 1, 0x40490FDB)])]
    blob2 = DT.serialise(multi)
    out2, dump2 = host.dta_roundtrip(blob2)
    assert out2 == blob2 and dump2 == DT.dump(multi)


def test_dta_rejects_bad_images(host):
    from oracle import dta_tree as DT
    tree = DT.synth_tree(np.random.default_rng(1), target_bytes=600)
    blob = DT.serialise(tree)
    for cut in (0, 4, 6, 9, 40, len(blob) - 1):
        with pytest.raises(host.HostError) as e:
            host.dta_roundtrip(blob[:cut])
        assert e.value.code == 6  # eError_InvalidData
    bad = bytearray(blob)
    bad[9:13] = (77).to_bytes(4, "little")  # unknown child type (CDtaFile.cpp:503-504)
    with pytest.raises(host.HostError):
        host.dta_roundtrip(bytes(bad))
    zero = bytearray(blob)
    zero[5:7] = (0).to_bytes(2, "little")  # a tree with no children is invalid (CDtaFile.cpp:398-401)
    with pytest.raises(host.HostError):
        host.dta_roundtrip(bytes(zero))
