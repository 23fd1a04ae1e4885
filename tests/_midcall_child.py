"""Child process of tests/test_gpu_parity.py::test_gpu_lost_in_the_middle_of_a_call (real MI355X, libmodgpu_testing.so) and of
tests/san_lib_cases.py (the CPU stand-in runtime under ASan / UBSan / TSan): the GPU "fails" at a chosen piece and stage of a
call that has already begun, and the reference's contract -- Cycle cannot fail, CEncryptionCycler.cpp:4-14, callers unguarded at
CArk.cpp:338-339, 1135-1136, Modulate.cpp:485-486 -- has to hold: modgpu_cycle_auto_host and CEncryptionCycler::Cycle return the
reference's bytes, the strict entry points and MODGPU_REQUIRE_GPU=1 return the error.

    python tests/_midcall_child.py <MiB>[,<MiB>...] [--class] [--files DIR]

Needs MODGPU_MIN_GPU_BYTES below the sizes used.  Prints MIDCALL_OK <strict> at the end."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import modulate_amd as M  # noqa: E402
from oracle import oracle as O  # noqa: E402

M.use_testing_flavour()  # failure injection exists only in libmodgpu_testing.so (or the sanitizer build)
strict = M.gpu_required()
assert M.device_count() >= 1 and M.testing_hooks()
sizes = [int(float(x) * (1 << 20)) + 5 for x in sys.argv[1].split(",")]
via_class = "--class" in sys.argv
STAGES = (M.STAGE_FILL, M.STAGE_LAUNCH, M.STAGE_SYNC, M.STAGE_DRAIN, M.STAGE_AFTER_DRAIN)
PIECES = (0, M.INJECT_PIECE_MIDDLE, M.INJECT_PIECE_LAST)
if via_class:
    from modulate_amd import host as H


def cycle(buf):
    """what an unmodified caller runs: Cycle(buf, n, key) through the class, or the entry point it binds to"""
    if via_class:
        H.cycle_via_class(buf, M.KEY_PS4)
    else:
        M.cycle_auto_host(buf, M.KEY_PS4)


for n in sizes:
    pt = O.splitmix_bytes(min(n, 64 << 20), 8)
    pt = np.resize(pt, n)
    want = pt.copy()
    O.cycle_at(want, O.KEY_PS4, 0)
    for piece in PIECES:
        for stage in STAGES:
            buf = pt.copy()  # pageable: what `new char[]` gives upstream's callers (CArk.cpp:320, 738, 780)
            before = M.path_stats()
            M.host_trace(True)
            M.debug_inject_failure_at(piece, stage)
            try:
                cycle(buf)
                assert not strict, "MODGPU_REQUIRE_GPU=1 must not compute on the host"
            except Exception as e:  # noqa: BLE001  (M.ModGpuError, or the host mirror's error when the class threw)
                assert strict, ("Cycle failed although the host loop was allowed", piece, stage, str(e))
            M.host_trace(False)
            assert not M.debug_injection_armed(), ("the failure never fired", n, piece, stage)
            after = M.path_stats()
            ev = M.host_trace_read()
            kinds = [e["kind"] for e in ev]
            assert "failed" in kinds, kinds[-8:]
            if strict:
                assert after["scalar_calls"] == before["scalar_calls"] == 0 and after["midcall_rescues"] == 0
                continue
            assert np.array_equal(buf, want), ("bytes differ from the reference's", n, piece, stage, int(np.flatnonzero(buf != want)[0]))
            assert after["midcall_rescues"] == before["midcall_rescues"] + 1, (before, after)
            assert after["auto_fallbacks"] == before["auto_fallbacks"] + 1
            rescued = after["midcall_rescued_bytes"] - before["midcall_rescued_bytes"]
            assert 0 <= rescued <= n and kinds[-2:] == ["rescued", "call_end"], (rescued, kinds[-4:])
            assert after["gpu_bytes"] - before["gpu_bytes"] == n - rescued
            # nothing was launched or filled for this call after every pipeline had seen the failure: the pipelines stop at their
            # next step, so at most one more fill / launch per pipeline follows the failure
            t_fail = next(i for i, k in enumerate(kinds) if k == "failed")
            late = sum(1 for k in kinds[t_fail:] if k in ("launched", "ready"))
            pipes = len({e["pipe"] for e in ev if e["pipe"] >= 0})
            assert late <= pipes, (late, pipes, n, piece, stage, [(e["kind"], e["pipe"], e["chunk"]) for e in ev[max(0, t_fail - 12):t_fail + 40]])
            if stage == M.STAGE_FILL and piece == 0:
                assert rescued > n // 2  # lost at the very start: the host loop does (nearly) everything
    if strict:
        continue
    # the GPU-only entry point never computes on the host: the error comes back (the buffer is then undefined -- documented)
    M.debug_inject_failure_at(M.INJECT_PIECE_MIDDLE, M.STAGE_SYNC)
    try:
        M.cycle_host(pt.copy(), M.KEY_PS4)
        raise SystemExit("modgpu_cycle_host computed after an injected failure")
    except M.ModGpuError as e:
        assert e.code == 3
    # no injection: the GPU serves the next call whole
    before = M.path_stats()
    buf = pt.copy()
    cycle(buf)
    assert np.array_equal(buf, want) and M.path_stats()["midcall_rescues"] == before["midcall_rescues"]

# ---- not the GPU but the HOST goes away: a pipeline thread stalls for four times the host-fed kernel's patience before it copies its piece
# in.  The kernel must give the call up by itself (every workgroup that waits for that chunk leaves, the others run out of tickets), the
# pipeline finds the kernel gone and the chunk undone, and the call ends like any other that lost its GPU under way.
import time  # noqa: E402
n = sizes[0]
pt = np.resize(O.splitmix_bytes(min(n, 64 << 20), 10), n)
want = pt.copy()
O.cycle_at(want, O.KEY_PS4, 0)
M.debug_set_host_tunable("feed_patience_ms", 40)
try:
    buf = pt.copy()
    before = M.path_stats()
    M.debug_inject_failure_at(M.INJECT_PIECE_MIDDLE, M.STAGE_STALL)
    t0 = time.perf_counter()
    try:
        cycle(buf)
        assert not strict, "MODGPU_REQUIRE_GPU=1 must not compute on the host"
    except Exception as e:  # noqa: BLE001
        assert strict and "gave up" in str(e), str(e)
    waited = time.perf_counter() - t0
    assert not M.debug_injection_armed() and 0.15 < waited < 20.0, waited
    if not strict:
        after = M.path_stats()
        assert np.array_equal(buf, want) and after["midcall_rescues"] == before["midcall_rescues"] + 1, (before, after)
    buf = pt.copy()  # and the next call finds a clean slate: no kernel left behind, counters at zero
    if strict:
        M.cycle_host(buf, M.KEY_PS4)
    else:
        cycle(buf)
    assert np.array_equal(buf, want) and M.last_launch()["variant"] == 4
finally:
    M.debug_set_host_tunable("feed_patience_ms", 10000)

if not strict:
    n = sizes[0]
    pt = np.resize(O.splitmix_bytes(min(n, 64 << 20), 9), n)
    want = pt.copy()
    O.cycle_at(want, O.KEY_PS4, 0)
    # ---- page-locked caller memory, cycled in place by ONE kernel across PCIe
    pb = M.PinnedBuffer(n)
    pb.array[:] = pt
    before = M.path_stats()
    M.debug_inject_failure_at(0, M.STAGE_LAUNCH)  # the launch fails: nothing has touched the pages -> the host loop does the buffer
    M.cycle_auto_host(pb.array, M.KEY_PS4)
    after = M.path_stats()
    assert np.array_equal(pb.array, want) and after["auto_fallbacks"] == before["auto_fallbacks"] + 1 and after["midcall_rescues"] == before["midcall_rescues"]
    pb.array[:] = pt
    M.debug_inject_failure_at(0, M.STAGE_SYNC)  # the kernel dies under way: nobody knows what it wrote -- the one case that stays an error
    try:
        M.cycle_auto_host(pb.array, M.KEY_PS4)
        raise SystemExit("the in-place route claimed success after its kernel had died")
    except M.ModGpuError as e:
        assert e.code == 3 and "page-locked" in str(e), str(e)
    pb.free()
    # ---- header-sized buffer through the kernel route (one slot, one launch): the caller's bytes change only after the wait succeeded
    small = pt[:300_000].copy()
    for stage in (M.STAGE_FILL, M.STAGE_LAUNCH, M.STAGE_SYNC):
        buf = small.copy()
        before = M.path_stats()
        M.debug_inject_failure_at(0, stage)
        M.cycle_auto_host(buf, M.KEY_PS4)
        w = small.copy()
        O.cycle_at(w, O.KEY_PS4, 0)
        assert np.array_equal(buf, w), ("header-sized buffer", stage, int(np.flatnonzero(buf != w)[0]), int((buf != w).sum()))
        assert M.path_stats()["auto_fallbacks"] == before["auto_fallbacks"] + 1 and not M.debug_injection_armed(), (stage, before, M.path_stats(), M.debug_injection_armed())
    # ---- modgpu_cycle_file_to_host (LoadArkData's part cipher): the file still holds every byte
    d = sys.argv[sys.argv.index("--files") + 1] if "--files" in sys.argv else tempfile.gettempdir()
    path = os.path.join(d, "midcall_%d.part" % os.getpid())
    pt.tofile(path)
    try:
        for pinned_dst in (False, True):
            for piece, stage in ((0, M.STAGE_FILL), (M.INJECT_PIECE_MIDDLE, M.STAGE_SYNC), (M.INJECT_PIECE_LAST, M.STAGE_LAUNCH), (M.INJECT_PIECE_MIDDLE, M.STAGE_AFTER_DRAIN)):
                if pinned_dst:
                    pb = M.PinnedBuffer(n)
                    dst = pb.array
                else:
                    dst = np.zeros(n, np.uint8)
                dst[:] = 0xEE
                before = M.path_stats()
                M.debug_inject_failure_at(piece, stage)
                M.cycle_file_to_host(path, n, M.KEY_PS4, out=dst)
                after = M.path_stats()
                assert not M.debug_injection_armed()
                assert np.array_equal(dst, want), ("file -> host", pinned_dst, piece, stage)
                assert after["midcall_rescues"] == before["midcall_rescues"] + 1 and after["auto_fallbacks"] == before["auto_fallbacks"]
                if pinned_dst:
                    pb.free()
        # ---- the other two file routes: host -> file (SaveArk's part cipher; the source memory is never modified) and file -> file,
        # to another file and in place (a piece is written only when it is finished, so the unwritten ones are still plaintext)
        out_path = path + ".out"
        for pinned_src in (False, True):
            if pinned_src:
                pb = M.PinnedBuffer(n)
                pb.array[:] = pt
                src = pb.array
            else:
                src = pt.copy()
            for piece, stage in ((0, M.STAGE_LAUNCH), (M.INJECT_PIECE_MIDDLE, M.STAGE_SYNC), (M.INJECT_PIECE_LAST, M.STAGE_DRAIN)):
                before = M.path_stats()
                M.debug_inject_failure_at(piece, stage)
                M.cycle_host_to_file(src, out_path, M.KEY_PS4)
                assert not M.debug_injection_armed() and M.path_stats()["midcall_rescues"] == before["midcall_rescues"] + 1
                assert np.array_equal(np.fromfile(out_path, dtype=np.uint8), want), ("host -> file", pinned_src, piece, stage)
                assert np.array_equal(src, pt)
            if pinned_src:
                pb.free()
        for in_place in (False, True):
            for piece, stage in ((0, M.STAGE_FILL), (M.INJECT_PIECE_MIDDLE, M.STAGE_SYNC), (M.INJECT_PIECE_LAST, M.STAGE_AFTER_DRAIN)):
                pt.tofile(path)
                before = M.path_stats()
                M.debug_inject_failure_at(piece, stage)
                M.cycle_file(path, path if in_place else out_path, M.KEY_PS4)
                assert not M.debug_injection_armed() and M.path_stats()["midcall_rescues"] == before["midcall_rescues"] + 1
                assert np.array_equal(np.fromfile(path if in_place else out_path, dtype=np.uint8), want), ("file -> file", in_place, piece, stage)
        os.unlink(out_path)
    finally:
        os.unlink(path)
print("MIDCALL_OK", strict)
sys.exit(0)
