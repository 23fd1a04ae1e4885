"""Cases for the sanitizer builds of libmodgpu's OWN host code (`make -C modulate_amd/csrc sanitize-lib`).

Not collected by a plain `pytest tests/`: tests/test_sanitizers.py runs this file in a child process with
MODGPU_LIB pointing at _san/libmodgpu_asan.so or _san/libmodgpu_tsan.so and the matching runtime preloaded.
Those builds link tests/cpu_runtime_standin/ instead of the HIP runtime: streams are real threads, a "launch" executes the launch
plan with the product's own arithmetic on the CPU, eight devices exist.  What runs under the sanitizers is
therefore exactly the code that cannot be sanitized on the GPU: launch planning, the ticket ring, the host-range
table, the staging pipelines' retire / refill state machine and their error paths, per-device contexts, worker
threads, per-thread error strings.  Every case also compares bytes with the oracle (the plan is checked too)."""
import ctypes
import os
import threading

import numpy as np
import pytest

import modulate_amd as M
from oracle import oracle as O

pytestmark = pytest.mark.skipif(not os.environ.get("MODGPU_LIB"), reason="runs only against a sanitizer build (tests/test_sanitizers.py)")


@pytest.fixture(scope="module")
def lib():
    L = M.lib()
    assert M.testing_hooks() and M.device_count() == 8, "expects the shim build with MODGPU_SHIM_DEVICES=8"
    L.modgpu_shim_pair_collisions.restype = ctypes.c_ulonglong
    L.modgpu_shim_launches.restype = ctypes.c_ulonglong
    L.modgpu_shim_batch_launches.restype = ctypes.c_ulonglong
    L.modgpu_shim_batch_plan_errors.restype = ctypes.c_ulonglong
    L.modgpu_shim_feed_launches.restype = ctypes.c_ulonglong
    L.modgpu_shim_feed_gave_up.restype = ctypes.c_ulonglong
    return L


@pytest.fixture()
def hooks(lib):
    with M.testing_flavour():
        try:
            yield M
        finally:
            M.debug_set_launch(None, 0)
            M.debug_set_pinned_mode(0)
            M.debug_set_staged_mode(0)
            M.debug_set_queue_ring(0)
            M.debug_set_helpers(0)
            M.debug_set_batch(0)
            M.debug_inject_failures(0)
            M.debug_inject_failure_at(0, -1)
            M.debug_set_gpu_node(-2)
            M.debug_set_host_tunable("feed", 1)
            M.debug_set_host_tunable("file_feed", 1)
            M.debug_set_host_tunable("feed_chunk_bytes", 256 << 10)
            M.debug_forbid_worker_threads(False)


def want(pt, key, off=0):
    w = pt.copy()
    O.cycle_at(w, key, off)
    return w


def test_launch_plan_sizes_alignments_shapes(hooks):
    """head / body / tail split, alignment lead and jump-ahead states of every launch shape."""
    cap = 400_000
    d = M.DeviceBuffer(cap + 256)
    rng = np.random.default_rng(1)
    for shape in (None, "small", "large", "queue"):
        M.debug_set_launch(shape, 3 if shape else 0)
        for n in (0, 1, 15, 16, 17, 4095, 4096, 4097, 65536 - 16, 65536 + 16, 131072 + 5, 300_001):
            for al in (0, 1, 4, 15):
                key = [0x90CFC0AB, 0xC64EED30, 1, 0, 0x7FFFFFFF][(n + al) % 5]
                so = [0, O.PERIOD - 7, (1 << 40) + 3][(n + al) % 3]
                whole = rng.integers(0, 256, size=n + 128, dtype=np.uint8)
                d.upload(whole)
                d.cycle(key, n=n, offset=64 + al, stream_off=so)
                d.sync()
                w = whole.copy()
                O.cycle_at(w[64 + al:64 + al + n], key, so)
                assert np.array_equal(d.download(n + 128), w), (shape, n, al)
    d.free()


def test_batch_plan_of_several_parts_in_one_launch(hooks, lib):
    """modgpu_cycle_batch_device / _parts_device: the plan of a launch that carries several parts -- every part its own
    split, lead, base states and slice of the chunk index space (the shim checks start[] tiles it) -- for ragged sizes
    (empty, edge-only, inside the cut first chunk, chunk multiples), every alignment class, own stream offsets, more
    than sixteen parts (two launches), and the one-launch-per-part fallbacks (mode 2; ring busy)."""
    rng = np.random.default_rng(7)
    sizes = [0, 1, 15, 16, 17, 31, 4096, 65535, 65536, 65537, 131072, 200_003, 3, 70_000, 0, 65536 * 3 + 9, 12, 40_000, 65536 * 2]
    slack = 96
    cap = sum(sizes) + slack * len(sizes) + 65536
    d = M.DeviceBuffer(cap)
    for mode, ring in ((1, 0), (2, 0), (0, 0)):
        M.debug_set_batch(mode)
        before, launches0 = M.queue_stats(), lib.modgpu_shim_batch_launches()
        whole = rng.integers(0, 256, size=cap, dtype=np.uint8)
        d.upload(whole)
        w = whole.copy()
        ptrs, offs, pos = [], [], 13
        for i, n in enumerate(sizes):
            pos += (i * 7) % 16                      # every alignment class against the 16-byte word
            ptrs.append(d.ptr + pos)
            offs.append([0, O.PERIOD - 5, (1 << 33) + i][i % 3])
            O.cycle_at(w[pos:pos + n], 0xC64EED30, offs[-1])
            pos += n + slack
        M.cycle_batch_device(ptrs, sizes, 0xC64EED30, stream_offs=offs, device=d.device)
        d.sync()
        assert np.array_equal(d.download(cap), w), mode
        after = M.queue_stats()
        carried = after["batch_parts"] - before["batch_parts"]
        if mode != 2:  # 17 non-empty parts: sixteen in one launch; the seventeenth is alone in its run: its own launch
            assert after["batch_launches"] - before["batch_launches"] == 1 and carried == 16, after  # (mode 0: small on average)
            assert lib.modgpu_shim_batch_launches() - launches0 == 1
            assert M.last_launch()["variant"] != 3
        else:
            assert carried == 0
    # the shipped rule says no to a few mid-sized parts that together fit the Infinity Cache
    M.debug_set_batch(0)
    two = [M.DeviceBuffer(26 << 20, device=1) for _ in range(2)]
    before = M.queue_stats()
    M.cycle_batch_device([b.ptr for b in two], [b.nbytes for b in two], 0, device=1)   # (key 0: identity, nothing launched)
    M.cycle_batch_device([b.ptr for b in two], [64, 64], 5, device=1)
    M.cycle_batch_device([b.ptr for b in two], [b.nbytes for b in two], 5, device=1)
    two[0].sync()
    after = M.queue_stats()
    assert after["batch_launches"] - before["batch_launches"] == 1 and after["batch_parts"] - before["batch_parts"] == 2
    for b in two:
        b.free()
    # without stream offsets (every part from 0), through the per-device grouping of modgpu_cycle_parts_device
    M.debug_set_batch(1)
    bufs = [M.DeviceBuffer(n, device=dev) for n, dev in ((70_001, 2), (5, 2), (300_000, 5), (65536, 2), (99_999, 5))]
    pts = [rng.integers(0, 256, size=b.nbytes, dtype=np.uint8) for b in bufs]
    for b, pt in zip(bufs, pts):
        b.upload(pt)
    before = M.queue_stats()
    M.cycle_parts_device(bufs, 0x90CFC0AB)
    after = M.queue_stats()
    assert after["batch_launches"] - before["batch_launches"] == 2 and after["batch_parts"] - before["batch_parts"] == 5
    for b, pt in zip(bufs, pts):
        assert np.array_equal(b.download(b.nbytes), want(pt, 0x90CFC0AB))
    # ring of one line, held by a launch in flight on another stream: the batch falls back to one launch per part
    M.debug_set_queue_ring(1)
    M.debug_set_launch("queue", 0)
    big = M.DeviceBuffer(1 << 20, device=2)                # (the ring is per device)
    big.upload(np.zeros(1 << 20, np.uint8))
    before = M.queue_stats()
    big.cycle(1)                                       # takes the only line (the shim's launches last >= 200 us)
    M.cycle_batch_device([b.ptr for b in bufs[:2]], [b.nbytes for b in bufs[:2]], 0x90CFC0AB, device=2)
    for b in bufs[:2]:
        b.sync()
    big.sync()
    after = M.queue_stats()
    assert after["batch_launches"] == before["batch_launches"] and after["busy_fallbacks"] > before["busy_fallbacks"]
    for b, pt in zip(bufs[:2], pts[:2]):
        assert np.array_equal(b.download(b.nbytes), pt)  # cycled twice: back to the plaintext
    assert lib.modgpu_shim_batch_plan_errors() == 0 and lib.modgpu_shim_pair_collisions() == 0
    for b in bufs + [big, d]:
        b.free()


def test_staged_pipelines_and_routes(hooks):
    """Pageable buffers through 8 pipelines x 2 slots (MODGPU_HOST_CHUNK_MB=1: many chunks on few MiB), both forms of
    the staged route, pinned caller memory through both pinned routes, registered memory, placed memory."""
    assert M.host_tunables()["chunk_bytes"] == 1 << 20
    ck = M.host_chunking()
    assert ck["ramp_bytes"] == 256 << 10 and ck["lanes"] == 4  # the 17 MiB case below takes the ramped plan, kernels on four shared lanes
    big_off = (1 << 64) - 70000  # (a staged buffer's chunk positions are added to the offset: the sum must not wrap at 2^64)
    z = np.zeros((5 << 20) + 3, np.uint8)
    assert np.array_equal(M.cycle_host(z.copy(), M.KEY_PS3, stream_off=big_off), want(z, M.KEY_PS3, big_off))
    for n in (1, 4097, (1 << 20) + 1, (3 << 20) - 1, (17 << 20) + 5):
        pt = O.splitmix_bytes(n + 16, n)
        for mode in (0, 1, 2):
            M.debug_set_staged_mode(mode)
            buf = pt.copy()
            M.cycle_host(buf[9:9 + n], M.KEY_PS3, stream_off=77)
            w = pt.copy()
            O.cycle_at(w[9:9 + n], M.KEY_PS3, 77)
            assert np.array_equal(buf, w), (n, mode)
    M.debug_set_staged_mode(0)
    n = (5 << 20) + 3
    pt = O.splitmix_bytes(n, 5)
    for pb in (M.PinnedBuffer(n + 64), M.PinnedBuffer(n + 64, near_device=3), M.PinnedBuffer(n + 64, parts=[n // 2, n + 64 - n // 2], n_devices=8)):
        assert pb.pinned
        for mode in (1, 2):
            M.debug_set_pinned_mode(mode)
            pb.array[:] = 0xEE
            pb.array[13:13 + n] = pt
            M.cycle_host(pb.array[13:13 + n], M.KEY_PS4, device=5)
            assert np.array_equal(pb.array[13:13 + n], want(pt, M.KEY_PS4)) and (pb.array[:13] == 0xEE).all() and (pb.array[13 + n:] == 0xEE).all()
        pb.free()
    M.debug_set_pinned_mode(0)
    buf = pt.copy()
    M.host_register(buf)
    M.cycle_host(buf, M.KEY_PS4)
    M.host_unregister(buf)
    assert np.array_equal(buf, want(pt, M.KEY_PS4))
    assert M.lib().modgpu_host_free(buf.ctypes.data) == 1 and M.lib().modgpu_host_unregister(buf.ctypes.data) == 1


def test_host_fed_kernel_route(hooks, lib):
    """Pageable memory on both sides: ONE host-fed kernel per call (cycle_feed_kernel.h) -- the stand-in runs the same ready / done /
    abort protocol on the stream's thread while the library's pipelines copy in and out.  Ragged ends (a short last piece, a tail that
    is not a whole word), one-chunk calls, 32 KiB chunks (many flags), stream offsets; the launch-per-chunk schedule when switched off."""
    before = lib.modgpu_shim_feed_launches()
    cases = [((1 << 20) + 1, 0), ((1 << 20) + 16, 7), ((2 << 20) + 4097, (1 << 40) + 3), ((3 << 20) - 1, 5), ((9 << 20) + 15, 0), (5 << 20, 123456789)]
    for chunk in (256 << 10, 32 << 10, 1 << 20):
        M.debug_set_host_tunable("feed_chunk_bytes", chunk)
        for n, off in cases:
            pt = O.splitmix_bytes(n + 32, n ^ chunk)
            buf = pt.copy()
            M.cycle_host(buf[7:7 + n], M.KEY_PS4, stream_off=off)
            w = pt.copy()
            O.cycle_at(w[7:7 + n], M.KEY_PS4, off)
            assert np.array_equal(buf, w), (chunk, n, off)
            assert M.last_launch()["variant"] == 4 and M.last_launch()["grid"] <= 32
    made = lib.modgpu_shim_feed_launches() - before
    assert made == 3 * len(cases) and lib.modgpu_shim_feed_gave_up() == 0
    M.debug_set_host_tunable("feed", 0)
    pt = O.splitmix_bytes((5 << 20) + 3, 3)
    assert np.array_equal(M.cycle_host(pt.copy(), M.KEY_PS4), want(pt, M.KEY_PS4))
    assert lib.modgpu_shim_feed_launches() - before == made  # a launch per chunk, as until round 5
    M.debug_set_host_tunable("feed", 1)


def test_no_worker_threads_to_be_had(hooks, lib, tmp_path):
    """the host-fed routes only work if a call's pipelines run side by side -- their one kernel draws chunks in stream order
    and waits for whichever pipeline owns the next.  When thread creation fails (a pids / NPROC limit) the caller runs the pipelines
    itself, one after another; such a call must take the launch-per-chunk schedule instead of parking 32 workgroups on chunks of
    pipelines that have not started.  Forbidding worker threads: same bytes, no host-fed launch, no pipeline run by a worker; allowed
    again: the host-fed kernel is back."""
    pt = O.splitmix_bytes((9 << 20) + 11, 77)
    path = tmp_path / "part.bin"
    pt.tofile(path)
    feeds, tasks = lib.modgpu_shim_feed_launches(), M.host_pool_stats()["pipelines_run_by_workers"]
    M.debug_forbid_worker_threads(True)
    try:
        assert np.array_equal(M.cycle_host(pt.copy(), M.KEY_PS4), want(pt, M.KEY_PS4))
        assert M.last_launch()["variant"] != 4
        pb = M.PinnedBuffer(pt.size)
        M.cycle_file_to_host(str(path), pt.size, M.KEY_PS4, out=pb.array)
        assert np.array_equal(pb.array, want(pt, M.KEY_PS4)) and M.last_launch()["variant"] != 4
        assert lib.modgpu_shim_feed_launches() == feeds and M.host_pool_stats()["pipelines_run_by_workers"] == tasks
        assert lib.modgpu_shim_feed_gave_up() == 0
    finally:
        M.debug_forbid_worker_threads(False)
    assert np.array_equal(M.cycle_host(pt.copy(), M.KEY_PS4), want(pt, M.KEY_PS4)) and M.last_launch()["variant"] == 4
    M.cycle_file_to_host(str(path), pt.size, M.KEY_PS4, out=pb.array)
    assert np.array_equal(pb.array, want(pt, M.KEY_PS4)) and M.last_launch()["variant"] == 4
    pb.free()
    assert lib.modgpu_shim_feed_launches() == feeds + 2


def test_concurrent_callers_share_one_device_and_the_parked_workers(hooks):
    """Round 4: a device's staging context hands SLOTS to calls instead of holding one mutex for a whole call, and its
    pipelines run on parked worker threads instead of threads spawned per call.  Four threads call modgpu_cycle_host on
    ONE device at once, pageable and pinned buffers mixed, several rounds: bytes exact, the calls really overlapped, no
    more than 15 workers were ever started for the device, and with 32 slots and 16 wanted per staged call the third
    and fourth caller had to take fewer pipelines or wait (both paths run).  The timeline hook records while this happens."""
    M.host_trace(True)
    before = M.host_pool_stats()
    sizes = [(5 << 20) + 3, (9 << 20) + 1, (3 << 20) + 7, (12 << 20) + 5]
    errors = []

    def caller(i):
        try:
            for r in range(3):
                pt = O.splitmix_bytes(sizes[i], 10 * i + r)
                got = M.cycle_host(pt.copy(), M.KEY_PS4, stream_off=i, device=2)
                assert np.array_equal(got, want(pt, M.KEY_PS4, i)), (i, r)
                small = O.splitmix_bytes(70_000 + i, r)  # the one-slot route in between
                assert np.array_equal(M.cycle_host(small.copy(), M.KEY_PS3, device=2), want(small, M.KEY_PS3))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=caller, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    M.host_trace(False)
    assert not errors, errors
    after = M.host_pool_stats()
    assert after["calls_overlapped"] > before["calls_overlapped"]
    assert after["pipelines_run_by_workers"] > before["pipelines_run_by_workers"]
    assert 0 < after["workers_started"] - before["workers_started"] <= 15
    ev = M.host_trace_read()
    kinds = {e["kind"] for e in ev}
    assert {"call_begin", "slots", "posted", "pipe_start", "fill_begin", "fill_end", "launched", "sync_begin", "sync_end", "drain_end",
            "pipe_end", "call_end"} <= kinds
    assert sum(e["kind"] == "call_begin" for e in ev) == sum(e["kind"] == "call_end" for e in ev) == 24
    assert all(b["t_ns"] >= a["t_ns"] for a, b in zip(ev, ev[1:]))  # recorded under one lock: monotonic


def test_header_sized_calls_do_not_wait_for_large_ones(hooks):
    """two large staged calls at the default 8 pipelines x 2 slots hold all 32 pipeline slots for their whole length; a
    header-sized Cycle on that GPU used to wait for one of them to END.  Two of a set's 34 slots are for one-slot calls only: with
    every pipeline slot held (the hook takes them the way a large call does), header-sized calls -- pageable through one slot, and
    page-locked memory in place -- are still served at once, two at a time."""
    assert M.host_pool_stats()["slots_per_device"] == 34
    assert M.debug_hold_slots(5, 64) == 32  # all a large call could ever get
    try:
        done = []

        def small(i):
            pt = O.splitmix_bytes(4092 + i, i)
            done.append(np.array_equal(M.cycle_host(pt.copy(), M.KEY_PS4, device=5), want(pt, M.KEY_PS4)))
            pb = M.PinnedBuffer(300_000, device=5) if False else M.PinnedBuffer(300_000)
            pb.array[:] = 9
            M.cycle_host(pb.array, M.KEY_PS3, device=5)
            done.append(np.array_equal(pb.array, want(np.full(300_000, 9, np.uint8), M.KEY_PS3)))
            pb.free()

        ts = [threading.Thread(target=small, args=(i,), daemon=True) for i in range(3)]
        [t.start() for t in ts]
        [t.join(timeout=60) for t in ts]
        assert not any(t.is_alive() for t in ts), "a header-sized call is waiting for slots that only large calls should compete for"
        assert done == [True] * 6
    finally:
        assert M.debug_hold_slots(5, 0) == 0
    big = O.splitmix_bytes((9 << 20) + 1, 4)  # and the pipeline slots are back
    assert np.array_equal(M.cycle_host(big.copy(), M.KEY_PS4, device=5), want(big, M.KEY_PS4))


def test_file_routes_and_their_error_paths(hooks, tmp_path):
    for n in (0, 1, (2 << 20) + 3, (9 << 20) + 11):
        pt = O.splitmix_bytes(n, n + 3)
        src, dst = tmp_path / f"p{n}", tmp_path / f"c{n}"
        pt.tofile(src)
        M.cycle_file(src, dst, M.KEY_PS4)
        assert np.array_equal(np.fromfile(dst, dtype=np.uint8), want(pt, M.KEY_PS4) if n else pt)
        M.cycle_file(dst, dst, M.KEY_PS4)
        assert np.array_equal(np.fromfile(dst, dtype=np.uint8), pt)
        assert np.array_equal(M.cycle_file_to_host(src, n, M.KEY_PS4), want(pt, M.KEY_PS4) if n else pt)
        pb = M.PinnedBuffer(n + 1)
        M.cycle_file_to_host(src, n, M.KEY_PS4, out=pb.array[:n])
        M.cycle_host_to_file(pb.array[:n], dst, M.KEY_PS4)
        assert np.array_equal(np.fromfile(dst, dtype=np.uint8), pt)
        pb.free()
    for call in (lambda: M.cycle_file(tmp_path / "missing", tmp_path / "x", M.KEY_PS4),
                 lambda: M.cycle_file_to_host(tmp_path / f"p{(9 << 20) + 11}", (9 << 20) + 99, M.KEY_PS4),  # short file: workers drain and stop
                 lambda: M.cycle_host_to_file(np.zeros(10, np.uint8), tmp_path / "no" / "such" / "dir", M.KEY_PS4)):
        with pytest.raises(M.ModGpuError) as e:
            call()
        assert e.value.code == 5 and str(e.value)


def test_eight_workers_and_per_thread_errors(hooks):
    sizes = [0, 1, 4096, 1_000_003, (3 << 20) + 5, 77, (2 << 20) + 1, 0, 500_000, 16, (1 << 20) - 3]
    for n_dev in (8, 3, 0):
        parts = [O.splitmix_bytes(s, 100 + i) for i, s in enumerate(sizes)]
        ws = [want(p, M.KEY_PS3) for p in parts]
        M.cycle_parts_host(parts, M.KEY_PS3, n_dev)
        assert all(np.array_equal(g, w) for g, w in zip(parts, ws)), n_dev
    # one host buffer split over the GPUs, every span its own stream offset (modgpu_cycle_host_split).  Spans are at least
    # 64 MiB, so this is 130 MiB through the byte-by-byte stand-in: under ASan only (the threads involved are the ones the
    # part lists above already ran under TSan)
    if "tsan" not in os.path.basename(os.environ["MODGPU_LIB"]):
        big = O.splitmix_bytes((130 << 20) + 77, 5)
        off = O.PERIOD - (70 << 20)  # two spans of 66 MiB: the period wraps inside the second
        got = big.copy()
        M.cycle_host_split(got, M.KEY_PS3, off, 8)
        assert np.array_equal(got, want(big, M.KEY_PS3, off))
    bufs = [M.DeviceBuffer(200_000 + 16 * i, device=i) for i in range(8)]
    for b in bufs:
        b.upload(np.zeros(b.nbytes, np.uint8))
    M.cycle_parts_device(bufs, M.KEY_PS4)
    for b in bufs:
        assert np.array_equal(b.download(), O.keystream(M.KEY_PS4, b.nbytes))
        b.free()
    # every thread keeps its own error text; the counters are shared
    errs = {}

    def worker(i):
        try:
            M.cycle_host(np.zeros(10, np.uint8), M.KEY_PS4, device=8 + i)
        except M.ModGpuError as e:
            errs[i] = (e.code, str(e))
        pt = O.splitmix_bytes(300_000 + i, i)
        assert np.array_equal(M.cycle_host(pt.copy(), M.KEY_PS4, device=i), want(pt, M.KEY_PS4))
        M.path_stats()

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert len(errs) == 8 and all(c == 1 and "out of range" in s for c, s in errs.values())


def test_injected_failures_and_the_second_engine(hooks):
    assert not M.gpu_required()
    pt = O.splitmix_bytes(3_000_001, 8)
    M.path_stats(reset=True)
    M.debug_inject_failures(2)
    with pytest.raises(M.ModGpuError) as e:
        M.cycle_host(pt.copy(), M.KEY_PS4)  # the GPU-only entry point never falls back
    assert e.value.code == 3
    buf = pt.copy()
    M.cycle_auto_host(buf, M.KEY_PS4)  # injected failure -> the host loop finishes the call
    assert np.array_equal(buf, want(pt, M.KEY_PS4))
    M.cycle_auto_host(buf, M.KEY_PS4)  # no injection left: the (shim) GPU serves it
    assert np.array_equal(buf, pt)
    small = O.splitmix_bytes(4092, 2)
    assert np.array_equal(M.cycle_auto_host(small.copy(), M.KEY_PS4), want(small, M.KEY_PS4))
    st = M.path_stats()
    assert st["auto_fallbacks"] == 1 and st["auto_small"] == 1 and st["scalar_calls"] == 2 and st["gpu_calls"] == 1, st


def test_gpu_lost_in_the_middle_of_a_call(hooks, tmp_path):
    """a failure injected at piece 0 / the middle piece / the last, at every stage (fill, launch, sync, drain, after
    the drain), of a staged call of 8 pipelines: the sibling pipelines stop, the host loop finishes exactly the pieces that have
    not arrived, the bytes are the reference's; the strict entry point returns the error; page-locked memory in place keeps its
    error once the kernel runs; modgpu_cycle_file_to_host reads the lost pieces again.  (tests/_midcall_child.py, in this process:
    the same script the GPU suite runs against the real runtime.)"""
    import runpy
    import sys
    argv = sys.argv
    sys.argv = ["_midcall_child.py", "12,33", "--files", str(tmp_path)]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_midcall_child.py"), run_name="__main__")
        assert e.value.code == 0
    finally:
        sys.argv = argv
        M.debug_inject_failure_at(0, -1)


def test_host_goes_away_under_a_waiting_kernel(hooks):
    """Only the host-stall case of tests/_midcall_child.py: tests/test_sanitizers.py runs THIS one again with the stand-in "GPU" slowed ten
    times (MODGPU_SHIM_SLOW=10), to show that the case no longer depends on who reaches the stalled chunk first."""
    import runpy
    import sys
    argv = sys.argv
    sys.argv = ["_midcall_child.py", "12", "--only-stall"]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_midcall_child.py"), run_name="__main__")
        assert e.value.code == 0
    finally:
        sys.argv = argv
        M.debug_inject_failure_at(0, -1)


def test_staging_set_of_another_numa_node(hooks):
    """Round 5: a call whose pageable pages live on another NUMA node than the GPU's takes that node's staging set -- slots from
    the library's own placed mapping (reserve, mbind, touch, hipHostRegister) instead of hipHostMalloc, workers bound to the node.
    This machine may have one node; the library is told its GPU hangs off node 1, so that node 0's pages are "elsewhere".  Parity
    on sizes that grow the slots (free + re-place), a failure injected in the middle of such a call, then back to the GPU's set."""
    before = M.host_pool_stats()["calls_on_another_nodes_set"]
    M.debug_set_gpu_node(1)
    try:
        for n in ((3 << 20) + 1, (17 << 20) + 5, (5 << 20) - 3):
            pt = O.splitmix_bytes(n, n)
            assert np.array_equal(M.cycle_host(pt.copy(), M.KEY_PS4), want(pt, M.KEY_PS4)), n
        moved = M.host_pool_stats()["calls_on_another_nodes_set"] - before
        if moved == 0:
            pytest.skip("the kernel would not say which node a page is on (get_mempolicy refused): nothing to place")
        assert moved == 3
        pt = O.splitmix_bytes((12 << 20) + 5, 77)
        buf = pt.copy()
        M.debug_inject_failure_at(M.INJECT_PIECE_MIDDLE, M.STAGE_SYNC)
        M.cycle_auto_host(buf, M.KEY_PS4)
        assert np.array_equal(buf, want(pt, M.KEY_PS4)) and not M.debug_injection_armed()
        # two callers at once on that set
        out = {}

        def run(i):
            p = O.splitmix_bytes((9 << 20) + i, i)
            out[i] = np.array_equal(M.cycle_host(p.copy(), M.KEY_PS3, device=0), want(p, M.KEY_PS3))
        ts = [threading.Thread(target=run, args=(i,)) for i in range(3)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert out == {0: True, 1: True, 2: True}
    finally:
        M.debug_set_gpu_node(-2)
    at = M.host_pool_stats()["calls_on_another_nodes_set"]
    pt = O.splitmix_bytes((4 << 20) + 9, 5)
    assert np.array_equal(M.cycle_host(pt.copy(), M.KEY_PS4), want(pt, M.KEY_PS4))
    assert M.host_pool_stats()["calls_on_another_nodes_set"] == at  # the GPU's own set again


def test_ticket_ring_under_concurrent_streams(hooks, lib):
    """Two host threads, each with its own stream, launching queue-shape work at the same time.  Ring of ONE line: the
    launches must never hold the pair together (the shim counts a launch that finds its pair taken)."""
    for ring in (1, 2, 0):
        M.debug_set_queue_ring(ring)
        M.debug_set_launch("queue", 4)
        before = lib.modgpu_shim_pair_collisions()
        q0 = M.queue_stats()
        n = 200_000
        out = {}

        def run(i):
            st = ctypes.c_void_p()
            assert lib.modgpu_shim_stream_create(ctypes.byref(st)) == 0
            pt = O.splitmix_bytes(n + 32, 50 + i)
            d = M.DeviceBuffer(n + 32)
            d.upload(pt)
            for _ in range(41):
                d.cycle(M.KEY_PS4, n=n, offset=7 + i, stream_off=i, stream=st.value)
            d.sync(stream=st.value)
            w = pt.copy()
            O.cycle_at(w[7 + i:7 + i + n], M.KEY_PS4, i)
            out[i] = np.array_equal(d.download(), w)
            d.free()
            assert lib.modgpu_shim_stream_destroy(st) == 0

        ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        q1 = M.queue_stats()
        assert out == {0: True, 1: True}, ring
        assert lib.modgpu_shim_pair_collisions() == before, "a ticket pair was handed to two overlapping launches"
        assert q1["eager"] + q1["busy_fallbacks"] - q0["eager"] - q0["busy_fallbacks"] == 82
        if ring == 1:
            assert q1["busy_fallbacks"] > q0["busy_fallbacks"]


def test_calls_leave_the_callers_device_alone(hooks, lib):
    """an entry point called with an explicit device must not leave the calling thread's current HIP device
    switched (in a torch process that would silently move torch's work to another GPU)."""
    assert lib.modgpu_shim_set_device(5) == 0
    pt = O.splitmix_bytes(100_000, 1)
    M.cycle_host(pt.copy(), M.KEY_PS4, device=2)
    M.cycle_host(O.splitmix_bytes((3 << 20) + 1, 2), M.KEY_PS4, device=7)  # worker threads
    d = M.DeviceBuffer(70_000, device=3)
    d.upload(pt[:70_000])
    d.cycle(M.KEY_PS4)
    d.sync()
    M.cycle_parts_device([d], M.KEY_PS4)
    assert np.array_equal(d.download(), pt[:70_000])
    d.free()
    M.cycle_parts_host([pt.copy(), pt.copy(), pt.copy()], M.KEY_PS4, 3)
    assert lib.modgpu_shim_get_device() == 5
    b = M.DeviceBuffer(16)  # device=-1: the calling thread's current device is the one used
    b.free()
    assert lib.modgpu_shim_get_device() == 5
    assert lib.modgpu_shim_set_device(0) == 0


def test_zz_a_host_fed_kernel_that_stops_responding_costs_the_device_not_the_process(hooks, lib):
    """LAST in this file (it costs device 7 its host-buffer routes for the rest of the process).  A host-fed kernel that neither
    finishes its chunks nor ends -- a workgroup wedged at a barrier; what check_isa.py's EXEC rule keeps out of a build, and what a
    host must survive anyway (feed_wait would poll hipErrorNotReady for ever).  The pipelines give the call up after the
    host-side deadline (4 x the kernel's patience + a grace period), nobody waits unboundedly on the kernel's stream, the call is
    finished by the host loop -- Cycle's bytes are the reference's --, the device's routes are abandoned (its slots are never handed
    out again), the next Cycle on it is served by the host loop, the GPU-only entry point says why, and the other devices work on."""
    import time
    lib.modgpu_shim_wedge_next_feed.argtypes = [ctypes.c_int]
    pt = O.splitmix_bytes((12 << 20) + 5, 123)
    w = want(pt, M.KEY_PS4)
    M.debug_set_host_tunable("feed_patience_ms", 300)  # deadline 4 x 0.3 s + 3 s of grace (a patience above any scheduling hiccup under the sanitizers)
    try:
        before = M.path_stats()
        lib.modgpu_shim_wedge_next_feed(1)
        t0 = time.perf_counter()
        buf = pt.copy()
        M.cycle_auto_host(buf, M.KEY_PS4, device=7)
        took = time.perf_counter() - t0
        after = M.path_stats()
        assert np.array_equal(buf, w) and after["midcall_rescues"] == before["midcall_rescues"] + 1, (before, after)
        assert 4.0 < took < 60.0, took  # it waited for the deadline and the grace period, and for nothing else
        # the device is gone for host buffers: Cycle falls to the host loop before it touches anything, the strict entry point explains
        buf = pt.copy()
        M.cycle_auto_host(buf, M.KEY_PS4, device=7)
        assert np.array_equal(buf, w) and M.path_stats()["auto_fallbacks"] == after["auto_fallbacks"] + 1
        with pytest.raises(M.ModGpuError) as e:
            M.cycle_host(pt.copy(), M.KEY_PS4, device=7)
        assert e.value.code == 3 and "abandoned" in str(e.value)
        small = pt[:100_000].copy()
        with pytest.raises(M.ModGpuError):
            M.cycle_host(small, M.KEY_PS4, device=7)  # the one-slot route too
        # its neighbours are untouched
        assert np.array_equal(M.cycle_host(pt.copy(), M.KEY_PS4, device=6), w) and M.last_launch()["variant"] == 4
    finally:
        M.debug_set_host_tunable("feed_patience_ms", 10000)
        lib.modgpu_shim_release_wedged()
