"""CPU-only checks of the product library: it loads, exports every symbol its two headers declare,
its host-side jump-ahead arithmetic agrees with the oracle, the GPU entry points fail loudly when
there is no GPU, and the library's own host loop (what `CEncryptionCycler::Cycle` falls back to on a
GPU-less host) reproduces the reference's golden vectors."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(modgpu_[a-z0-9_]+)\s*\(", src))


def test_header_symbols_all_exported(modgpu):
    assert _declared("modgpu.h") == set(modgpu.EXPORTS), _declared("modgpu.h") ^ set(modgpu.EXPORTS)
    hooks = set(modgpu.TESTING_EXPORTS) | set(modgpu.DEBUG_EXPORTS)
    assert _declared("modgpu_testing.h") == hooks, _declared("modgpu_testing.h") ^ hooks
    assert all(n.startswith("modgpu_debug_") for n in modgpu.DEBUG_EXPORTS)
    assert not any(n.startswith("modgpu_debug_") for n in list(modgpu.EXPORTS) + list(modgpu.TESTING_EXPORTS))
    L = modgpu.lib()
    assert modgpu.active_flavour() == "shipped" and not modgpu.testing_hooks()
    for name in list(modgpu.EXPORTS) + list(modgpu.TESTING_EXPORTS):
        assert getattr(L, name) is not None
    assert L.modgpu_abi_version() == 8

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("modgpu_")}

    # The shipped library exports exactly the ABI + the reporting hooks: no knob that changes how it launches or makes a
    # call fail exists in it.  The testing flavour is the same plus the modgpu_debug_* hooks.
    shipped = exported(modgpu.lib_path("shipped"))
    assert shipped == set(modgpu.EXPORTS) | set(modgpu.TESTING_EXPORTS), shipped ^ (set(modgpu.EXPORTS) | set(modgpu.TESTING_EXPORTS))
    assert not [n for n in shipped if "debug" in n]
    assert exported(modgpu.lib_path("testing")) == shipped | set(modgpu.DEBUG_EXPORTS)
    with pytest.raises(modgpu.ModGpuError):
        modgpu.debug_set_launch("queue", 4)  # outside testing_flavour(): refused, the shipped library has no such hook
    with modgpu.testing_flavour():
        assert modgpu.testing_hooks() and modgpu.active_flavour() == "testing"
        assert modgpu.kernel_source_hash() == _shipped_hash(modgpu)  # same device code in both
        modgpu.debug_set_launch(None, 0)
    assert modgpu.active_flavour() == "shipped"


def _shipped_hash(modgpu):
    return modgpu.capi._load("shipped").modgpu_kernel_source_hash().decode()


def test_kernel_source_hash_matches_sources(modgpu):
    import hashlib
    h = hashlib.sha256()
    for f in ("cycle_kernel_impl.h", "cycle_kernel.hip", "cycle_kernel.h", "lcg.h"):
        h.update(open(os.path.join(ROOT, "modulate_amd", "csrc", f), "rb").read())
    assert modgpu.kernel_source_hash() == h.hexdigest()
    # the host-fed kernel's TU has an identity of its own: modgpu_last_launch reports it for variant 4 and the
    # roofline_pcie profiles record it
    h = hashlib.sha256()
    for f in ("cycle_feed_kernel.hip", "cycle_feed_kernel.h", "cycle_kernel_impl.h", "lcg.h"):
        h.update(open(os.path.join(ROOT, "modulate_amd", "csrc", f), "rb").read())
    assert modgpu.feed_kernel_source_hash() == h.hexdigest() != modgpu.kernel_source_hash()


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


def test_gpu_entry_points_fail_loudly_without_gpu(modgpu):
    """Without a GPU the GPU entry points must fail, not quietly compute on the host."""
    if modgpu.device_count() > 0:
        pytest.skip("GPU present")
    buf = np.arange(64, dtype=np.uint8)
    keep = buf.copy()
    with pytest.raises(modgpu.ModGpuError) as e:
        modgpu.cycle_host(buf, modgpu.KEY_PS4)
    assert e.value.code == 2 and np.array_equal(buf, keep)
    hdr = np.zeros(64, np.uint8)
    hdr[:4] = np.frombuffer(modgpu.MAGIC_PS4.to_bytes(4, "little"), np.uint8)
    if modgpu.gpu_required():  # the framing entry points follow Cycle's dispatch: only strict mode keeps them on the GPU
        with pytest.raises(modgpu.ModGpuError):
            modgpu.hdr_decrypt_host(hdr)
    with pytest.raises(modgpu.ModGpuError):
        modgpu.cycle_parts_host([buf], modgpu.KEY_PS4)
    with pytest.raises(modgpu.ModGpuError) as e:
        modgpu.cycle_batch_device([buf.ctypes.data], [buf.size], modgpu.KEY_PS4, device=0)
    assert e.value.code == 2 and np.array_equal(buf, keep)
    with pytest.raises(modgpu.ModGpuError) as e:
        modgpu.DeviceBuffer(64)
    assert e.value.code == 2
    st = modgpu.path_stats()
    assert st["gpu_calls"] == 0 and st["gpu_launches"] == 0


def test_missing_extension_fails_loudly(tmp_path):
    """No library, no result: the Python layer has no arithmetic of its own to fall back to (the oracle is never imported
    by the package), so a missing libmodgpu.so is an error at the first call."""
    code = ("import numpy as np, modulate_amd as M\n"
            "try:\n"
            "    M.cycle_auto_host(np.zeros(64, np.uint8), M.KEY_PS4)\n"
            "except M.ModGpuError as e:\n"
            "    print('LOUD', e.code, 'not built' in str(e))\n")
    e = dict(os.environ, PYTHONPATH=ROOT, MODGPU_LIB=str(tmp_path / "no_such_libmodgpu.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=120)
    assert r.returncode == 0 and "LOUD -1 True" in r.stdout, r.stdout + r.stderr[-1500:]
    for name in os.listdir(os.path.join(ROOT, "modulate_amd")):  # and the package never reaches into oracle/
        if name.endswith(".py"):
            src = open(os.path.join(ROOT, "modulate_amd", name)).read()
            assert "import oracle" not in src and "from oracle" not in src, name


def test_argument_errors(modgpu):
    bad = np.zeros(64, np.uint8)
    with pytest.raises(modgpu.ModGpuError) as e:
        modgpu.hdr_decrypt_host(bad)
    # magic is checked before any device work (CArk.cpp:328-334): code 4 even without a GPU
    assert e.value.code == 4
    with pytest.raises(modgpu.ModGpuError) as e:
        modgpu.hdr_decrypt_host(np.zeros(3, np.uint8))
    assert e.value.code == 1


# ---- the library's own host loop (modgpu_cycle_scalar_host / _auto_host) -----------------------
needs_host_loop = pytest.mark.skipif(os.environ.get("MODGPU_REQUIRE_GPU", "0") not in ("", "0"),
                                     reason="MODGPU_REQUIRE_GPU forbids the host loop in this process")


@needs_host_loop
def test_host_loop_golden_vectors(modgpu, oracle, golden):
    """Every fixture the compiled reference produced (tests/golden): keystreams for the 16 keys incl.
    INT_MIN / -1 / zero-residue, the 34 plaintext cases, the SURVEY 4 KiB vector -- through the product's
    host loop, on the build box."""
    for e in golden["keystream"]:
        ks = modgpu.cycle_scalar_host(np.zeros(1 << 20, np.uint8), e["key"])
        assert ks[:64].tobytes().hex() == e["first64"], hex(e["key"])
        assert f"{oracle.fnv1a64(ks):016x}" == e["fnv_1m"]
    for e in golden["plaintext_cases"]:
        pt = oracle.splitmix_bytes(e["n"], e["seed"])
        ct = modgpu.cycle_scalar_host(pt.copy(), e["key"])
        assert f"{oracle.fnv1a64(ct):016x}" == e["ct_fnv"], e
        assert ct[:16].tobytes().hex() == e["ct_first16"] and ct[-16:].tobytes().hex() == e["ct_last16"]
        assert np.array_equal(modgpu.cycle_scalar_host(ct, e["key"]), pt)
    b = ((np.arange(4096, dtype=np.uint32) * 131 + 7) & 0xFF).astype(np.uint8)
    ct = modgpu.cycle_scalar_host(b.copy(), golden["survey_4k"]["key"])
    assert f"{oracle.fnv1a64(ct):016x}" == golden["survey_4k"]["ct_fnv"]
    # the reference's 2^32-1-byte run: samples around the period wrap and at the very end, by stream offset
    L = golden["large"]
    for smp in list(L["samples"]) + [{"off": L["around_period"]["start"], "hex": L["around_period"]["hex"]},
                                     {"off": L["tail16"]["start"], "hex": L["tail16"]["hex"]}]:
        m = len(smp["hex"]) // 2
        assert modgpu.cycle_scalar_host(np.zeros(m, np.uint8), L["key"], stream_off=smp["off"]).tobytes().hex() == smp["hex"]


@needs_host_loop
def test_host_loop_matches_oracle_sizes_offsets_threads(modgpu, oracle):
    keys = [0x90CFC0AB, 0xC64EED30, 1, 0xFFFFFFFF, 0x80000000, 0, 0x7FFFFFFF, 0x80000001, 12345]
    offs = [0, 5, 16, oracle.PERIOD - 3, oracle.PERIOD, (1 << 32) - 1, (1 << 40) + 7, (1 << 64) - 70000]
    for n in (0, 1, 15, 16, 17, 31, 4092, 100_001):
        for key in keys:
            for off in offs:
                pt = oracle.splitmix_bytes(n + 8, n + 1)
                got = pt.copy()
                modgpu.cycle_scalar_host(got[3:3 + n], key, off)  # misaligned view, guard bytes either side
                want = pt.copy()
                oracle.cycle_at(want[3:3 + n], key, off)
                assert np.array_equal(got, want), (n, hex(key), off)
    # >= 4 MiB: contiguous spans on several host threads, each jumping to its own position
    for n, off in (((4 << 20) + 1, 0), ((33 << 20) + 77, oracle.PERIOD - (5 << 20))):
        pt = oracle.splitmix_bytes(n, n)
        want = pt.copy()
        oracle.cycle_at(want, 0xC64EED30, off)
        assert np.array_equal(modgpu.cycle_scalar_host(pt.copy(), 0xC64EED30, off), want), n


@needs_host_loop
def test_auto_entry_point_uses_host_loop_without_gpu(modgpu, oracle):
    """BASELINE config 1 as worded: a 4 KiB blob on the CPU path, no GPU -- through the entry point
    CEncryptionCycler::Cycle binds to.  Header-sized buffers are the host loop's by the size dispatch;
    larger ones end there only because no GPU can serve them."""
    if modgpu.device_count() > 0:
        pytest.skip("GPU present")
    assert modgpu.min_gpu_bytes() == 16 << 20  # the measured crossover against one host thread (profiles/r03_small_call_crossover.txt)
    before = modgpu.path_stats()
    body = oracle.splitmix_bytes(4092, 0x4D6F64756C617465)
    got = modgpu.cycle_auto_host(body.copy(), modgpu.KEY_PS4)
    assert np.array_equal(got, oracle.cycle(body.copy(), oracle.KEY_PS4))
    after = modgpu.path_stats()
    assert after["scalar_calls"] == before["scalar_calls"] + 1 and after["auto_small"] == before["auto_small"] + 1
    assert after["auto_fallbacks"] == before["auto_fallbacks"]
    assert after["scalar_bytes"] == before["scalar_bytes"] + 4092 and after["gpu_calls"] == before["gpu_calls"]
    # the framing of Cycle's call sites (CArk.cpp:328-339, 914-915) takes the same dispatch
    framed = np.concatenate([np.zeros(4, np.uint8), body])
    want = framed.copy()
    assert oracle.hdr_encrypt(want, True) == 0
    assert np.array_equal(modgpu.hdr_encrypt_host(framed, True), want)
    assert np.array_equal(modgpu.hdr_decrypt_host(framed)[4:], body)
    after = modgpu.path_stats()
    assert after["auto_small"] == before["auto_small"] + 3 and after["gpu_calls"] == before["gpu_calls"]
    big = oracle.splitmix_bytes((16 << 20) + 1, 5)
    assert np.array_equal(modgpu.cycle_auto_host(big.copy(), modgpu.KEY_PS3), oracle.cycle(big.copy(), oracle.KEY_PS3))
    last = modgpu.path_stats()
    assert last["auto_fallbacks"] == after["auto_fallbacks"] + 1 and last["auto_small"] == after["auto_small"]


@pytest.mark.parametrize("setting,want", [("0", 0), ("4096", 4096), ("0x100000", 1 << 20), ("junk", 16 << 20), ("", 16 << 20)])
def test_min_gpu_bytes_knob_is_latched_at_load(setting, want):
    """MODGPU_MIN_GPU_BYTES (SURVEY 5 'min-size-for-GPU knob'): read once; n < value -> host loop in modgpu_cycle_auto_host
    only.  On this GPU-less machine the route shows in the counters: below the threshold `auto_small`, at or above it
    `auto_fallbacks` (the GPU was tried first)."""
    code = ("import numpy as np, modulate_amd as M\n"
            "print('MIN', M.min_gpu_bytes())\n"
            "if M.device_count() == 0:\n"
            "    for n in (4095, 4096, 4097):\n"
            "        M.path_stats(reset=True); M.cycle_auto_host(np.zeros(n, np.uint8), M.KEY_PS4); st = M.path_stats()\n"
            "        assert st['scalar_calls'] == 1 and st['auto_small'] == (1 if n < M.min_gpu_bytes() else 0), (n, st)\n"
            "        assert st['auto_fallbacks'] == 1 - st['auto_small'], (n, st)\n"
            "print('KNOB_OK')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("MODGPU_MIN_GPU_BYTES", "MODGPU_REQUIRE_GPU")}
    env.update(MODGPU_MIN_GPU_BYTES=setting, PYTHONPATH=ROOT, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0 and "KNOB_OK" in r.stdout and f"MIN {want}\n" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("setting,want", [(None, "offload"), ("offload", "offload"), ("fastest", "fastest"), ("nonsense", "offload")])
def test_host_policy_is_latched_at_load_and_priced_from_the_table(setting, want):
    """what modgpu_cycle_auto_host does above MODGPU_MIN_GPU_BYTES is a stated, switchable policy.
    MODGPU_HOST_POLICY is read once; `fastest` decides per call from the committed crossover table
    (modulate_amd/csrc/crossover_table.h), pricing the host loop with the threads it would really get -- so with ONE thread
    allowed the kernel wins from the table's crossover up, with many threads the host loop wins on this node class, and
    page-locked memory (no staging copies) moves the kernel's price down."""
    code = ("import json, modulate_amd as M\n"
            "r = {'policy': M.host_policy(), 'min': M.min_gpu_bytes()}\n"
            "for mib in (1, 4, 16, 32, 64, 256, 1024):\n"
            "    r[str(mib)] = [M.host_policy_engine(mib << 20, p) for p in (False, True)]\n"
            "print('R', json.dumps(r))\n")
    import json
    out = {}
    for threads in ("1", "32"):
        env = {k: v for k, v in os.environ.items() if k not in ("MODGPU_HOST_POLICY", "MODGPU_HOST_THREADS")}
        env.update(PYTHONPATH=ROOT, MODGPU_HOST_THREADS=threads, MODGPU_HOST_ISA="avx512")
        if setting is not None:
            env["MODGPU_HOST_POLICY"] = setting
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout + r.stderr
        out[threads] = json.loads(r.stdout.split("R ", 1)[1])
        assert out[threads]["policy"] == want
    one, many = out["1"], out["32"]
    if "avx512" != __import__("modulate_amd").host_loop_isa():
        return  # (the prices below are the AVX-512 body's; a CPU without it is priced by its own body)
    # one host thread (17 GB/s): behind the staged kernel route from ~3 MiB up since round 5 (22 GB/s at 4 MiB; round 4: the two met
    # near 16 MiB), ahead at 1 MiB, where a call is still latency-bound
    assert one["1"][0][0] == "host" and one["4"][0][0] == "kernel"
    assert one["64"][0][0] == "kernel" and one["1024"][0][0] == "kernel"
    assert one["16"][1][0] == "kernel"  # page-locked memory: 50 GB/s across the link beats one thread at every size of the table
    # the threads this machine allows: priced at threads x 17.4 x 0.6 (crossover_table.h), so the table's answer depends on the machine's CPU count
    import multiprocessing
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else multiprocessing.cpu_count()
    if cpus >= 8:
        assert many["256"][0][0] == "host" and many["1024"][0][0] == "host"  # >= 8 threads: > 100 GB/s against a 50 GB/s link
    for mib in ("16", "64", "1024"):
        assert many[mib][0][1] <= one[mib][0][1]  # more threads never price the host loop slower
        assert one[mib][1][2] <= one[mib][0][2]   # pinned memory never prices the kernel slower than pageable


@needs_host_loop
def test_every_host_loop_body_matches_oracle_and_golden(modgpu, oracle, golden):
    """generic / avx2 / avx512 bodies of the host loop (scalar_path.cpp), each against the oracle over block
    boundaries of every body (16 / 32 / 64 bytes), misaligned views, period wrap and 64-bit offsets, and against
    the reference's own keystream digests."""
    ran = []
    for isa in ("generic", "avx2", "avx512"):
        probe = np.zeros(1, np.uint8)
        try:
            modgpu.cycle_scalar_host_isa(probe, 1, isa)
        except modgpu.ModGpuError as e:
            assert e.code == 1  # this CPU does not run that body
            continue
        ran.append(isa)
        for n in (0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 4092, 100_003):
            for key in (0x90CFC0AB, 0xC64EED30, 1, 0xFFFFFFFF, 0, 0x80000001):
                for off in (0, 7, oracle.PERIOD - 40, (1 << 32) - 1, (1 << 64) - 70000):
                    pt = oracle.splitmix_bytes(n + 8, n + 1)
                    got = pt.copy()
                    modgpu.cycle_scalar_host_isa(got[3:3 + n], key, isa, off)
                    want = pt.copy()
                    oracle.cycle_at(want[3:3 + n], key, off)
                    assert np.array_equal(got, want), (isa, n, hex(key), off)
        for ks in golden["keystream"]:
            z = np.zeros(1 << 20, np.uint8)
            modgpu.cycle_scalar_host_isa(z, ks["key"], isa)
            assert z[:64].tobytes().hex() == ks["first64"] and oracle.fnv1a64(z) == int(ks["fnv_1m"], 16), (isa, hex(ks["key"]))
    assert "generic" in ran and modgpu.host_loop_isa() == ran[-1]  # the automatic choice is the widest body the CPU runs
    with pytest.raises(modgpu.ModGpuError):
        modgpu.cycle_scalar_host_isa(np.zeros(4, np.uint8), 1, "sse9")


def test_numa_topology_reader_on_a_fake_sysfs(modgpu, tmp_path):
    """The sysfs reader behind modgpu_host_alloc_near / _alloc_parts and the worker placement, pointed at
    a fake two-socket tree."""
    sysfs = tmp_path / "sys"
    for bdf, node in (("0000:05:00.0", "0"), ("0000:85:00.0", "1"), ("0000:c5:00.0", "-1")):
        d = sysfs / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(node + "\n")
    for node, cpus in ((0, "0-3,128-131\n"), (1, "64-66,70,  200-201\n")):
        d = sysfs / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus)
    assert modgpu.numa_probe(str(sysfs), "0000:05:00.0") == (0, [0, 1, 2, 3, 128, 129, 130, 131])
    assert modgpu.numa_probe(str(sysfs), "0000:85:00.0") == (1, [64, 65, 66, 70, 200, 201])
    assert modgpu.numa_probe(str(sysfs), "0000:c5:00.0") == (-1, [])   # the kernel's own "no node"
    assert modgpu.numa_probe(str(sysfs), "0000:ff:00.0") == (-1, [])   # no such device
    assert modgpu.numa_probe(str(tmp_path / "nowhere"), "0000:05:00.0") == (-1, [])
    assert modgpu.numa_probe(str(sysfs), "0000:05:00.0", max_cpus=3) == (0, [0, 1, 2])


def test_placed_host_memory_without_gpu_is_plain_memory(modgpu):
    if modgpu.device_count() > 0:
        pytest.skip("GPU present")
    assert modgpu.device_numa_node(0) == -1
    for pb in (modgpu.PinnedBuffer(300_000, near_device=0), modgpu.PinnedBuffer(300_000, parts=[100_000, 0, 200_000], n_devices=8)):
        assert not pb.pinned
        pb.array[:] = 3
        assert int(pb.array.sum()) == 900_000
        pb.free()
    # the measurement hook that names the node outright: node 0 exists everywhere; a node that does not exist is refused or ignored
    import ctypes
    p = ctypes.c_void_p()
    L = modgpu.lib()
    assert L.modgpu_host_alloc_on_node(ctypes.byref(p), 1 << 20, 0) == 0 and p.value
    ctypes.memset(p, 0x5A, 1 << 20)
    assert L.modgpu_host_is_pinned(p, 1 << 20) == 0 and L.modgpu_host_free(p) == 0
    assert L.modgpu_host_alloc_on_node(ctypes.byref(p), 4096, -1) == 1


def test_require_gpu_forbids_the_host_loop(modgpu):
    """MODGPU_REQUIRE_GPU=1: no second engine.  (Child process: the switch is read once at load.)"""
    code = ("import numpy as np, modulate_amd as M\n"
            "assert M.gpu_required()\n"
            "b = np.arange(100, dtype=np.uint8); k = b.copy()\n"
            "for fn, want in ((lambda: M.cycle_scalar_host(b, M.KEY_PS4), (6,)), (lambda: M.cycle_auto_host(b, M.KEY_PS4), (2, 3))):\n"
            "    try:\n"
            "        fn(); raise SystemExit('computed on the host')\n"
            "    except M.ModGpuError as e:\n"
            "        assert e.code in want, e.code\n"
            "assert (b == k).all() and M.path_stats()['scalar_calls'] == 0\n"
            "print('FORBIDDEN_OK')\n")
    env = dict(os.environ, MODGPU_REQUIRE_GPU="1", HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0 and "FORBIDDEN_OK" in r.stdout, r.stdout + r.stderr


def test_host_alloc_without_gpu_is_plain_memory(modgpu):
    if modgpu.device_count() > 0:
        pytest.skip("GPU present")
    pb = modgpu.PinnedBuffer(100_000)
    assert not pb.pinned and pb.ptr % 64 == 0
    pb.array[:] = 7
    assert int(pb.array.sum()) == 700_000
    pb.free()
    assert modgpu.lib().modgpu_host_free(12345) == 1  # MODGPU_ERR_INVALID: not one of ours
    buf = np.arange(4096, dtype=np.uint8)
    modgpu.host_register(buf)  # no GPU: nothing to pin for, succeeds and changes nothing
    assert modgpu.lib().modgpu_host_is_pinned(buf.ctypes.data, buf.size) == 0
    modgpu.host_unregister(buf)


def test_device_alias_needs_a_device(modgpu):
    """MODGPU_DEVICE_ALIAS multiplies real devices; it never conjures one."""
    env = dict(os.environ, MODGPU_DEVICE_ALIAS="8", HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", "import modulate_amd as M; print('COUNT', M.device_count())"],
                       capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0 and "COUNT 0" in r.stdout, r.stdout + r.stderr


def test_kernel_codegen_guard_passes_the_tree_and_rejects_a_broken_build():
    """The streaming kernels' keystream is a hand-scheduled assembly block in FIXED registers; the work-queue kernel's ticket
    fetch must stay one plain returning atomic.  Whether the compiler kept to that is visible only in its output, so
    modulate_amd/csrc/check_isa.py reads the gfx950 assembly of BOTH kernel TUs -- `make` runs it before it will produce either
    object.  Here: the tree's TUs pass, and a build with an input of the block pinned into one of its fixed temporaries (round 3's
    wrong-keystream build) is REJECTED.  hipcc is part of the build container: its absence is a failure, not a skip."""
    hipcc = "/opt/rocm/bin/hipcc"
    assert os.path.exists(hipcc), "hipcc is missing: the code-generation guard cannot run, and that is not acceptable for a build box"
    csrc = os.path.join(ROOT, "modulate_amd", "csrc")
    flags = subprocess.run(["make", "-s", "-C", csrc, "--eval", "print-kflags: ; @echo $(KERNEL_FLAGS)", "print-kflags"],
                           capture_output=True, text=True).stdout.split()
    assert "-amdgpu-atomic-optimizer-strategy=None" in flags, flags
    good = subprocess.run(["make", "-s", "-C", csrc, "isa-check"], capture_output=True, text=True, timeout=900)
    assert good.returncode == 0 and "check_isa: ok (4 kernels)" in good.stdout, good.stdout[-3000:] + good.stderr[-2000:]
    broken = subprocess.run(["make", "-s", "-C", csrc, "isa-check-broken"], capture_output=True, text=True, timeout=900)
    assert broken.returncode != 0, "the guard accepted a build whose keystream block reads a register the block overwrites"
    assert "the compiler gave a block operand a fixed temporary" in broken.stdout, broken.stdout[-3000:]
    # the object file rules depend on the guard: a TU that fails it produces neither cycle_kernel.o nor cycle_feed_kernel.o
    mk = open(os.path.join(csrc, "Makefile")).read()
    stamp = mk[mk.index("isa_checked.stamp:"):mk.index("cycle_kernel.o:")]
    assert "check_isa.py cycle_kernel.s cycle_feed_kernel.s" in stamp
    for obj in ("cycle_kernel.o:", "cycle_feed_kernel.o:"):
        assert "isa_checked.stamp" in mk[mk.index(obj):].splitlines()[0], obj


def _check_isa():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "modulate_amd", "csrc", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_codegen_guard_sdwa_forwarding_rule():
    """on gfx940+ a VALU that reads a VGPR directly behind an SDWA write of PART of it (dst_sel BYTE_n) needs a
    wait state; LLVM's hazard recognizer does not look into inline assembly, and the small shape's and the host-fed kernel's put_byte
    is one such instruction per asm statement.  Until round 6 only the scheduler's habit of putting a shift between two of them kept
    that safe; now check_isa.py refuses a TU in which any kernel has such a pair.  The tree passes (previous test); a build whose
    put_byte is followed at once by a reader of the dword is rejected, in both TUs; and so is the tree's own assembly with the one
    instruction of margin taken out."""
    csrc = os.path.join(ROOT, "modulate_amd", "csrc")
    broken = subprocess.run(["make", "-s", "-C", csrc, "isa-check-broken-sdwa"], capture_output=True, text=True, timeout=900)
    assert broken.returncode != 0, "the guard accepted a read directly behind an SDWA partial write"
    out = broken.stdout
    assert "dst_sel forwarding hazard" in out and "modgpu_cycle_feed_kernel" in out and "modgpu_cycle_kernelILi1ELi256E" in out, out[-3000:]
    ci = _check_isa()
    subprocess.check_call(["make", "-s", "-C", csrc, "cycle_feed_kernel.s"])
    asm = open(os.path.join(csrc, "cycle_feed_kernel.s")).read()
    assert ci.check(asm) == []
    lines = asm.splitlines()
    sd = [i for i, ln in enumerate(lines) if "v_add_u32_sdwa" in ln and "UNUSED_PRESERVE" in ln]
    hit = 0
    for a, b in zip(sd, sd[1:]):  # two consecutive put_byte's of ONE dword: drop what the scheduler put between them
        if lines[a].split()[1] == lines[b].split()[1]:
            twin = "\n".join(lines[:a + 1] + lines[b:])
            assert any("dst_sel forwarding hazard" in f for f in ci.check(twin)), (lines[a], lines[b])
            hit += 1
    assert hit >= 1  # (if the scheduler ever stops interleaving, the tree itself fails the guard -- which is the point)


def test_codegen_guard_barriers_need_the_whole_wave():
    """VERDICT r5 #2(ii) / ADVICE: the host-fed kernel's lab form hung a workgroup -- lanes 1..63 of one wave went round the trip
    loop's back edge without lane 0 and met the barrier twice per trip.  It was fixed by source shape; check_isa.py now follows the
    COMPILED control flow of every kernel with a stack of saved EXEC masks and refuses a build in which an s_barrier can be reached
    with part of the wave masked off.  Here: the tree's four kernels pass that rule (8 barriers between them), and the host-fed
    kernel's assembly with the EXEC restore in front of a barrier taken out -- the wave arrives with only thread 0's region's mask --
    is rejected; so are nt stores and a third barrier."""
    import re
    csrc = os.path.join(ROOT, "modulate_amd", "csrc")
    ci = _check_isa()
    subprocess.check_call(["make", "-s", "-C", csrc, "cycle_kernel.s", "cycle_feed_kernel.s"])
    main = open(os.path.join(csrc, "cycle_kernel.s")).read()
    feed = open(os.path.join(csrc, "cycle_feed_kernel.s")).read()
    for asm in (main, feed):
        for name, fn in ci.kernel_texts(asm).items():
            assert ci.barriers_at_full_exec(name, fn) == [], name
    assert sum(fn.count("s_barrier") for asm in (main, feed) for fn in ci.kernel_texts(asm).values()) >= 8
    # every exec restore that directly precedes a barrier, taken out in turn
    lines = feed.splitlines()
    twins = 0
    for i, ln in enumerate(lines):
        if ln.strip() == "s_barrier":
            back = [k for k in range(max(0, i - 4), i) if lines[k].strip().startswith("s_or_b64 exec, exec,")]
            if back:
                twin = "\n".join(lines[:back[-1]] + lines[back[-1] + 1:])
                assert any("part of the wave masked off" in f for f in ci.check(twin)), lines[back[-1]]
                twins += 1
    assert twins == 2
    assert any("not `sc1` without nt" in f for f in ci.check(re.sub(r"(buffer_store_dwordx4 [^\n]*) sc1", r"\1 nt sc1", feed)))
    assert any("s_barrier, expected 2" in f for f in ci.check(feed.replace("\ts_barrier\n", "\ts_barrier\n\ts_barrier\n", 1)))
    md = ci.metadata(feed, next(iter(ci.kernel_bodies(feed))))
    assert md["vgpr_count"] <= 64 and md["vgpr_spill_count"] == 0 and md["sgpr_spill_count"] == 0 and md["private_segment_fixed_size"] == 0


def test_lab_kernel_at_the_products_settings_is_the_products_loop():
    """tools/cycle_kernel_lab.h repeats the work-queue kernel with its tuning knobs (VERDICT r3 #8 moved them out of the product
    header).  A copy can drift.  This compiles tools/tune_cycle.hip to gfx950 assembly and compares the PRODUCT instantiation
    modgpu_cycle_queue_kernel<4, 1024> with the lab kernel at the product's settings, opcode by opcode: the same instructions the
    same number of times, give or take the wait / hazard-nop bookkeeping that moves with the kernel-argument layout.  So an A/B row
    of tools/tune_cycle against "the lab form" is an A/B against what ships."""
    import collections
    hipcc = "/opt/rocm/bin/hipcc"
    assert os.path.exists(hipcc), "hipcc is missing"
    tools, csrc = os.path.join(ROOT, "tools"), os.path.join(ROOT, "modulate_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", tools, "golden_kat.inc"])
    flags = subprocess.run(["make", "-s", "-C", csrc, "--eval", "print-kflags: ; @echo $(KERNEL_FLAGS)", "print-kflags"], capture_output=True, text=True).stdout.split()
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-I" + csrc, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                        os.path.join(tools, "tune_cycle.hip"), "-o", "-"], capture_output=True, text=True, timeout=900, cwd=tools)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = r.stdout

    def histogram(label):
        m = re.search(r"^(" + label + r"):", asm, re.M)
        assert m, label
        h = collections.Counter()
        for ln in asm[m.end():asm.index("s_endpgm", m.end())].splitlines():
            ln = ln.strip()
            if ln and not ln.startswith((";", ".")) and not ln.endswith(":"):
                h[ln.split()[0]] += 1
        return h
    product = histogram(r"_Z25modgpu_cycle_queue_kernelILi4ELi1024EEv14CycleQueueArgs")
    # <U 4, BLOCK 1024, ALG 2, SAUX sc1|nt, TRACE 0, DEPTH 1, MODE_FULL, LAUX nt, B1 1, B2 1, TSPLIT 0, TK 1, TLOOP 0, LSP 0, HSB 0>
    lab = histogram(r"_Z22lab_cycle_queue_kernelILi4ELi1024ELi2ELi18ELi0ELi1ELi0ELi2ELi1ELi1ELi0ELi1ELi0ELi0ELi0EEv12LabQueueArgs")
    bookkeeping = {"s_waitcnt", "s_nop", "s_mov_b32", "s_mov_b64"}
    diff = {k: (product[k], lab[k]) for k in set(product) | set(lab) if product[k] != lab[k]}
    assert all(k in bookkeeping and abs(a - b) <= 4 for k, (a, b) in diff.items()), diff
    assert product["v_mad_u64_u32"] == lab["v_mad_u64_u32"] > 250 and product["buffer_load_dwordx4"] == lab["buffer_load_dwordx4"] == 13
    assert product["global_atomic_add"] == lab["global_atomic_add"] == 4 and product["s_barrier"] == lab["s_barrier"]


@pytest.mark.parametrize("env,want", [
    ({}, {"pipes": 8, "chunk_bytes": 8 << 20, "zerocopy_max_bytes": 1 << 20, "ring": 4}),
    # a zero-copy limit above the slot size used to overrun the pinned staging slot
    ({"MODGPU_HOST_CHUNK_MB": "1", "MODGPU_HOST_ZEROCOPY_KB": "2048"}, {"chunk_bytes": 1 << 20, "zerocopy_max_bytes": 1 << 20}),
    ({"MODGPU_HOST_ZEROCOPY_KB": "999999999"}, {"zerocopy_max_bytes": 8 << 20}),
    ({"MODGPU_HOST_PIPES": "99", "MODGPU_HOST_CHUNK_MB": "0", "MODGPU_HOST_RING": "9", "MODGPU_HOST_ZEROCOPY_KB": "0"},
     {"pipes": 16, "chunk_bytes": 1 << 20, "ring": 4, "zerocopy_max_bytes": 0}),
])
def test_host_tunables_are_clamped_at_load(env, want):
    # (MODGPU_HOST_ZEROCOPY_KB and _RING are knobs of the testing flavour only since round 5; _PIPES and _CHUNK_MB of both)
    code = "import json, modulate_amd as M; M.use_testing_flavour(); print('T', json.dumps(M.host_tunables()))"
    e = {k: v for k, v in os.environ.items() if not k.startswith("MODGPU_HOST_")}
    e.update(env, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    got = json.loads(r.stdout.split("T ", 1)[1])
    for k, v in want.items():
        assert got[k] == v, (k, got)
    assert got["zerocopy_max_bytes"] <= got["chunk_bytes"]


def test_host_loop_pool_survives_fork():
    """the parked workers of the host loop do not exist in a fork()ed child (Python's multiprocessing forks by default),
    but the pool used to count them still: the child ran every span on its one thread, its request queue grew without bound, and
    a fork taken while a worker held the pool's mutex deadlocked the child's first threaded call.  The child gets a fresh pool."""
    code = r"""
import os, sys, numpy as np
import modulate_amd as M
from oracle import oracle as O
pt = O.splitmix_bytes(48 << 20, 3)
want = O.cycle_at(pt.copy(), M.KEY_PS4, 0)
info = lambda: (lambda o: (M.lib().modgpu_host_loop_info(o), list(o))[1])((__import__("ctypes").c_uint64 * 4)())
assert np.array_equal(M.cycle_scalar_host(pt.copy(), M.KEY_PS4), want)
started = info()[3]
pid = os.fork()
if pid == 0:
    ok = np.array_equal(M.cycle_scalar_host(pt.copy(), M.KEY_PS4), want) and np.array_equal(M.cycle_scalar_host(pt.copy(), M.KEY_PS4), want)
    grew = info()[3] - started  # the child had to start workers of its own (it inherited the count, not the threads)
    os._exit(0 if ok and (grew > 0 or started == 0) else 3)
_, status = os.waitpid(pid, 0)
assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
assert np.array_equal(M.cycle_scalar_host(pt.copy(), M.KEY_PS4), want)  # the parent's pool is as it was
print("FORK_OK", started)
"""
    e = dict(os.environ, PYTHONPATH=ROOT, MODGPU_REQUIRE_GPU="0", MODGPU_HOST_THREADS="4")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, cwd=ROOT, timeout=300)
    assert r.returncode == 0 and "FORK_OK" in r.stdout, r.stdout + r.stderr
    assert int(r.stdout.split("FORK_OK")[1]) >= 1  # (the parent did run threaded: the case is real)


def test_environment_of_the_shipped_library_is_what_the_header_lists():
    """the variables libmodgpu.so reads are exactly the ones include/modgpu.h documents (ten); the staging and
    host-loop measurement knobs exist in the testing flavour only, and the shipped library ignores them."""
    import re
    def names(path):
        out = subprocess.run(["strings", "-a", path], capture_output=True, text=True, check=True).stdout
        return {w for w in out.split() if re.fullmatch(r"MODGPU_[A-Z0-9_]+", w) and not w.startswith(("MODGPU_ERR", "MODGPU_ISA", "MODGPU_KERNEL", "MODGPU_TESTING"))}
    shipped = names(os.path.join(ROOT, "modulate_amd", "libmodgpu.so"))
    header = open(os.path.join(ROOT, "include", "modgpu.h")).read()
    block = header[header.index(" * Environment (each read once"):header.index("#ifndef MODGPU_H")]
    listed = set(re.findall(r"^ \*     (MODGPU_[A-Z0-9_]+)", block, flags=re.M))
    assert shipped == listed == {"MODGPU_REQUIRE_GPU", "MODGPU_MIN_GPU_BYTES", "MODGPU_HOST_POLICY", "MODGPU_HOST_THREADS", "MODGPU_HOST_ISA", "MODGPU_DEVICE_ALIAS",
                                 "MODGPU_NUMA", "MODGPU_HELPER_BELOW_MHZ", "MODGPU_HOST_PIPES", "MODGPU_HOST_CHUNK_MB"}, (sorted(shipped), sorted(listed))
    testing = names(os.path.join(ROOT, "modulate_amd", "libmodgpu_testing.so"))
    assert testing - shipped == {"MODGPU_HOST_ZEROCOPY_KB", "MODGPU_HOST_RING", "MODGPU_HOST_SPLIT", "MODGPU_HOST_CHUNK_MIN_MB", "MODGPU_HOST_RAMP_KB", "MODGPU_HOST_LANES",
                                  "MODGPU_HOST_NTCOPY", "MODGPU_HOST_FILE_SCHED", "MODGPU_HOST_FILE_FEED", "MODGPU_HOST_FEED", "MODGPU_HOST_FEED_CHUNK_KB", "MODGPU_HOST_SPREAD", "MODGPU_HOST_CGROUP"}, sorted(testing - shipped)
    code = ("import json, modulate_amd as M; a = [M.host_tunables(), M.host_chunking()]; M.use_testing_flavour(); "
            "print('T', json.dumps([a, [M.host_tunables(), M.host_chunking()]]))")
    e = {k: v for k, v in os.environ.items() if not k.startswith("MODGPU_HOST_")}
    e.update(PYTHONPATH=ROOT, MODGPU_HOST_ZEROCOPY_KB="64", MODGPU_HOST_LANES="1", MODGPU_HOST_RAMP_KB="0", MODGPU_HOST_SPLIT="99", MODGPU_HOST_RING="2")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    (st, sc), (tt, tc) = json.loads(r.stdout.split("T ", 1)[1])
    assert st["zerocopy_max_bytes"] == 1 << 20 and st["ring"] == 4 and sc["lanes"] == 4 and sc["ramp_bytes"] == 512 << 10 and sc["split"] == 64 and sc["chunk_min_bytes"] == 1 << 20, (st, sc)
    assert tt["zerocopy_max_bytes"] == 64 << 10 and tt["ring"] == 2 and tc["lanes"] == 1 and tc["ramp_bytes"] == 0 and tc["split"] == 99, (tt, tc)


def test_headers_are_plain_c_and_link_standalone(tmp_path):
    """include/*.h compile as C99 with -pedantic -Werror; a C program linked against libmodgpu.so alone reproduces
    the reference's PS4 keystream vector through the entry point Cycle binds to (tests/c/abi_smoke.c)."""
    exe = str(tmp_path / "abi_smoke")
    lib = os.path.join(ROOT, "modulate_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L" + lib, "-lmodgpu", "-Wl,-rpath," + lib])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and ("ABI_SMOKE_OK" in r.stdout or os.environ.get("MODGPU_REQUIRE_GPU", "0") not in ("", "0")), r.stdout + r.stderr


def test_codegen_guard_exec_tracking_on_hand_written_snippets():
    """The barrier rule's control-flow walk on assembly small enough to read: a barrier behind the EXEC restore passes; inside a
    saveexec region, inside a loop whose lanes leave one by one (s_andn2 exec), or reachable through a branch that skips the restore,
    it is refused; a region that is closed on both paths of an if / else passes."""
    ci = _check_isa()
    ok = """
        s_and_saveexec_b64 s[4:5], vcc
        s_cbranch_execz .LBB0_2
        v_mov_b32 v0, 1
    .LBB0_2:
        s_or_b64 exec, exec, s[4:5]
        s_barrier
        s_endpgm
    """
    assert ci.barriers_at_full_exec("ok", ok) == []
    inside = ok.replace("        v_mov_b32 v0, 1\n", "        v_mov_b32 v0, 1\n        s_barrier\n")
    assert any("masked off" in f for f in ci.barriers_at_full_exec("inside", inside))
    skipped = """
        s_and_saveexec_b64 s[4:5], vcc
        s_cbranch_execz .LBB0_3
        v_mov_b32 v0, 1
        s_or_b64 exec, exec, s[4:5]
    .LBB0_3:
        s_barrier
        s_endpgm
    """  # the execz path jumps over the restore: the wave can meet the barrier with the narrowed mask
    assert any("masked off" in f for f in ci.barriers_at_full_exec("skipped", skipped))
    lane_loop = """
    .LBB0_1:
        v_cmp_eq_u32_e32 vcc, 0, v1
        s_or_b64 s[6:7], vcc, s[6:7]
        s_andn2_b64 exec, exec, s[6:7]
        s_barrier
        s_cbranch_execnz .LBB0_1
        s_or_b64 exec, exec, s[6:7]
        s_endpgm
    """
    assert any("masked off" in f for f in ci.barriers_at_full_exec("lane_loop", lane_loop))
    lane_loop_ok = lane_loop.replace("        s_barrier\n", "").replace("        s_endpgm", "        s_barrier\n        s_endpgm")
    assert ci.barriers_at_full_exec("lane_loop_ok", lane_loop_ok) == []
    if_else = """
        s_and_saveexec_b64 s[4:5], vcc
        s_xor_b64 s[4:5], exec, s[4:5]
        s_cbranch_execz .LBB0_2
        v_mov_b32 v0, 1
    .LBB0_2:
        s_or_saveexec_b64 s[4:5], s[4:5]
        s_xor_b64 exec, exec, s[4:5]
        s_cbranch_execz .LBB0_4
        v_mov_b32 v0, 2
    .LBB0_4:
        s_or_b64 exec, exec, s[4:5]
        s_barrier
        s_endpgm
    """
    assert ci.barriers_at_full_exec("if_else", if_else) == []
    # the SDWA rule on a snippet: one instruction of margin is enough, none is not; a store of the register counts as a read
    sd = "        v_add_u32_sdwa v5, v1, v2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
    assert ci.sdwa_forwarding_hazards("a", sd + "        v_xor_b32_e32 v6, v5, v7\n        s_endpgm\n")
    assert ci.sdwa_forwarding_hazards("b", sd + "        global_store_dword v[8:9], v5, off\n        s_endpgm\n")
    assert not ci.sdwa_forwarding_hazards("c", sd + "        s_nop 0\n        v_xor_b32_e32 v6, v5, v7\n        s_endpgm\n")
    assert not ci.sdwa_forwarding_hazards("d", sd + "        v_mov_b32_e32 v5, v7\n        s_endpgm\n")  # a whole overwrite is no read
    assert not ci.sdwa_forwarding_hazards("e", sd + "        v_xor_b32_e32 v6, v4, v7\n        s_endpgm\n")
