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
                        "-k", "header_writer or straddle or bad_arguments or config1 or framing_without_gpu or quirk or dta", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout
