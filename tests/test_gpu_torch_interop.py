"""GPU: the C ABI on memory and streams owned by PyTorch (plumbing only: torch supplies the device
buffer and the stream, the arithmetic is the HIP kernel), and hipGraph capture of the launch.

Runs in a child process that imports torch first: PyTorch bundles its own HIP runtime, and a
process must use one runtime (whichever of torch / libmodgpu.so loads first serves both)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_torch_memory_streams_and_graph_capture(modgpu):
    assert modgpu.device_count() >= 1
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_torch_interop_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TORCH_INTEROP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_every_state_in_every_byte_role(modgpu):
    """All 2^31 - 2 generator states at a word base in each kernel shape: see tests/_every_state_child.py."""
    assert modgpu.device_count() >= 1
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_every_state_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "EVERY_STATE_OK passes=52" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
