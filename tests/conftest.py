import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
