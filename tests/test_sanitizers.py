"""Runs the host mirror's CPU tests again against an ASan + UBSan build of libmodulate_host.so
(`make -C modulate_amd/csrc sanitize`).  CPU only: GPU sanitizers are not available on the pool.
The build links a stub of the C ABI whose GPU entry points report "no device" plus the product's own
host loop (scalar_path.cpp), so the host logic and that loop both execute under the sanitizers."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_logic_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "modulate_amd", "csrc"), "sanitize"])
    env = dict(os.environ,
               LD_PRELOAD=f"{asan}:{ubsan}",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",  # the Python interpreter itself "leaks"
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               MODULATE_HOST_LIB=os.path.join(ROOT, "modulate_amd", "_san", "libmodulate_host.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_cpu.py"), "-x", "-q",
                        "-k", "header_writer or straddle or bad_arguments or config1 or framing_without_gpu or quirk or dta or mutated", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout


def _san_lib_cases(preload, lib, extra_env, select=(), expect="16 passed"):
    from oracle import oracle as O
    O.build(ref=False)  # here, not in the child: the compiler must not run under a preloaded sanitizer runtime
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "modulate_amd", "csrc"), "sanitize-lib"])
    env = dict(os.environ, LD_PRELOAD=preload, MODGPU_LIB=os.path.join(ROOT, "modulate_amd", "_san", lib),
               MODGPU_SHIM_DEVICES="8", MODGPU_HOST_CHUNK_MB="1", MODGPU_HOST_RAMP_KB="256", MODGPU_REQUIRE_GPU="0", MODGPU_MIN_GPU_BYTES="65536", **extra_env)
    # (1 MiB slots and a 256 KiB ramp: buffers of 12 MiB and up take the ramped plan -- small first / last chunk per pipeline -- on few MiB)
    for k in ("MODGPU_HOST_PIPES", "MODGPU_HOST_ZEROCOPY_KB", "MODGPU_DEVICE_ALIAS", "MODGPU_HOST_SPLIT", "MODGPU_HOST_CHUNK_MIN_MB", "MODGPU_HOST_LANES"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "san_lib_cases.py"), "-x", "-q", "-p", "no:cacheprovider"] + list(select),
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=1500)
    assert r.returncode == 0 and expect in r.stdout, r.stdout[-4000:] + r.stderr[-4000:]


def test_library_host_code_under_asan_ubsan():
    """libmodgpu's own threaded host code (modgpu_capi.cpp, host_stream.cpp) built against tests/cpu_runtime_standin/ -- a CPU
    stand-in for the HIP runtime whose streams are real threads -- with ASan + UBSan, driven by tests/san_lib_cases.py:
    launch planning at every shape, staged pipelines and both route forms, pinned / registered / placed memory, file routes
    and their error paths, eight workers on eight devices, failure injection, the ticket ring under two concurrent streams."""
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    _san_lib_cases(f"{asan}:{ubsan}", "libmodgpu_asan.so",
                   {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})


def test_library_host_code_under_tsan():
    """The same cases under ThreadSanitizer: the staging pipelines' retire / refill state machine, the ring's bookkeeping, the
    host-range table and the shared counters, with worker threads and stream threads really running concurrently."""
    tsan = _runtime("libtsan.so")
    if not tsan:
        pytest.skip("gcc ThreadSanitizer runtime not installed")
    opts = {"TSAN_OPTIONS": f"halt_on_error=1 second_deadlock_stack=1 suppressions={os.path.join(ROOT, 'tests', 'tsan.supp')}"}
    _san_lib_cases(tsan, "libmodgpu_tsan.so", opts)


def test_midcall_cases_do_not_depend_on_who_arrives_first():
    """round 5's host-stall case slept for 4 x the kernel's patience from the moment the PIPELINE reached the
    chunk and demanded a rescue; under TSan on a loaded box the stand-in kernel reached the chunk when the stall was nearly over,
    waited less than its patience and finished the call normally -- a red test, one run in three, with a correct library.  The
    stall now lasts until the kernel has given up, whoever arrives first.  Proof: that case under TSan with the stand-in
    "GPU" slowed TEN times (MODGPU_SHIM_SLOW: every span it cycles takes ten times as long, so it is always the late one)."""
    tsan = _runtime("libtsan.so")
    if not tsan:
        pytest.skip("gcc ThreadSanitizer runtime not installed")
    opts = {"TSAN_OPTIONS": f"halt_on_error=1 second_deadlock_stack=1 suppressions={os.path.join(ROOT, 'tests', 'tsan.supp')}", "MODGPU_SHIM_SLOW": "10"}
    _san_lib_cases(tsan, "libmodgpu_tsan.so", opts, select=("-k", "host_goes_away_under_a_waiting_kernel"), expect="1 passed")
