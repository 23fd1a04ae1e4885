"""CPU-only checks of the product library: it loads, exports every symbol include/modgpu.h
declares, its host-side jump-ahead arithmetic agrees with the oracle, and compute calls fail
loudly (no CPU fallback) when there is no GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported(modgpu):
    src = open(os.path.join(ROOT, "include", "modgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = set(re.findall(r"\b(modgpu_[a-z0-9_]+)\s*\(", src))
    assert declared == set(modgpu.EXPORTS), declared ^ set(modgpu.EXPORTS)
    L = modgpu.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.modgpu_abi_version() == 2


def test_state_at_matches_oracle(modgpu, oracle):
    rng = np.random.default_rng(3)
    keys = [oracle.KEY_PS3, oracle.KEY_PS4, 0, 1, 0x7FFFFFFF, 0x80000000, 0x80000001, 0xFFFFFFFF, 12345]
    offs = [0, 1, 15, 16, 4095, 4096, oracle.PERIOD - 1, oracle.PERIOD, oracle.PERIOD + 1, (1 << 32) - 1, 1 << 32,
            (1 << 63) + 12345, (1 << 64) - 1] + [int(x) for x in rng.integers(0, 1 << 62, size=50)]
    for key in keys:
        for i in offs:
            assert modgpu.state_at(key, i) == oracle.state_at(key, i) & 0xFFFFFFFF, (hex(key), i)


def test_jump_tables(modgpu):
    m, a = 0x7FFFFFFF, 16807
    steps = {0: 1, 1: 16, 2: 4096, 3: 4096 * 256}
    sizes = {0: 16, 1: 256, 2: 256, 3: 256}
    for which, step in steps.items():
        t = modgpu.jump_table(which)
        assert len(t) == sizes[which]
        assert t == [pow(a, step * i, m) for i in range(len(t))]
    assert modgpu.jump_table(9) == []


def test_no_cpu_fallback(modgpu):
    """Without a GPU the compute entry points must fail, not quietly compute on the host."""
    if modgpu.device_count() > 0:
        pytest.skip("GPU present")
    buf = np.arange(64, dtype=np.uint8)
    keep = buf.copy()
    with pytest.raises(modgpu.ModGpuError) as e:
        modgpu.cycle_host(buf, modgpu.KEY_PS4)
    assert e.value.code == 2 and np.array_equal(buf, keep)
    hdr = np.zeros(64, np.uint8)
    hdr[:4] = np.frombuffer(modgpu.MAGIC_PS4.to_bytes(4, "little"), np.uint8)
    with pytest.raises(modgpu.ModGpuError):
        modgpu.hdr_decrypt_host(hdr)
    with pytest.raises(modgpu.ModGpuError):
        modgpu.cycle_parts_host([buf], modgpu.KEY_PS4)


def test_argument_errors(modgpu):
    bad = np.zeros(64, np.uint8)
    with pytest.raises(modgpu.ModGpuError) as e:
        modgpu.hdr_decrypt_host(bad)
    # magic is checked before any device work (CArk.cpp:328-334): code 4 even without a GPU
    assert e.value.code == 4
    with pytest.raises(modgpu.ModGpuError) as e:
        modgpu.hdr_decrypt_host(np.zeros(3, np.uint8))
    assert e.value.code == 1
