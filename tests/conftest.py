import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# On a machine with an AMD GPU (the kernel driver's /dev/kfd exists) every test process forbids the
# library's host loop BEFORE libmodgpu.so is loaded (the switch is read once at load): whatever a GPU
# test compares against the oracle was then computed by the gfx950 kernel or the call failed.
# tests/test_gpu_parity.py::test_engine_is_the_gpu checks the counters as well.
if os.path.exists("/dev/kfd"):
    os.environ.setdefault("MODGPU_REQUIRE_GPU", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "cycle_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.build(ref=False)
    return O


@pytest.fixture(scope="session")
def modgpu():
    """The product bindings; the HIP library must have been built (no fallback)."""
    import modulate_amd as M
    M.lib()
    return M


@pytest.fixture()
def header_cwd(tmp_path, monkeypatch):
    """A working directory that holds placeholder main_ps4.hdr / main_ps3.hdr files: the reference's
    SaveArk refuses to run unless it can open lpHeaderFilename there (Modulate/CArk.cpp:904-909), and
    so does this repo's by default."""
    d = tmp_path / "cwd"
    d.mkdir()
    for plat in ("ps3", "ps4"):
        (d / f"main_{plat}.hdr").write_bytes(b"")
    monkeypatch.chdir(d)
    return d
