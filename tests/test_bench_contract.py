"""CPU-only: bench.py refuses to run without the HIP path (it never measures a CPU stand-in), and its
multi-rank launch guard works."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=120)


def test_bench_fails_loudly_without_gpu(modgpu):
    if modgpu.device_count() > 0:
        import pytest
        pytest.skip("GPU present")
    r = _run(["--steps", "1", "--warmup", "0", "--part-bytes", "4096"])
    assert r.returncode != 0 and "no HIP device" in (r.stdout + r.stderr)
    assert '"metric"' not in r.stdout  # no JSON line is printed for a run that measured nothing


def test_bench_multi_rank_needs_launcher():
    r = _run(["--gpus", "2"])
    assert r.returncode != 0 and "torch.distributed.run" in (r.stdout + r.stderr)
