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


def test_traffic_is_replayed_only_for_the_same_device_code(modgpu, tmp_path):
    """`roofline.traffic` comes from the committed PMC summary, so it must drop to null when
    that summary was taken on other kernel sources, another instantiation or another part size."""
    import json
    code = ("import json, sys, bench\n"
            "import modulate_amd as M\n"
            "s = json.load(open(bench.ROOT + '/profiles/pmc_summary.json'))\n"
            "h = M.kernel_source_hash()\n"
            "k = 'modgpu_cycle_kernel<8, 1024, 2, true>'\n"
            "r = {'same': bench.load_traffic(int(s['part_bytes']), s['cycle_kernel'].split('(')[0].replace('void ', ''), s.get('kernel_source_hash')),\n"
            "     'other_hash': bench.load_traffic(int(s['part_bytes']), k, '0' * 64),\n"
            "     'other_size': bench.load_traffic(12345, k, s.get('kernel_source_hash')),\n"
            "     'other_kernel': bench.load_traffic(int(s['part_bytes']), 'modgpu_cycle_kernel<1, 256, 1, false>', s.get('kernel_source_hash')),\n"
            "     'loaded_matches': s.get('kernel_source_hash') == h}\n"
            "print('RESULT', json.dumps(r))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads(r.stdout.split("RESULT", 1)[1])
    for k in ("other_hash", "other_size", "other_kernel"):
        assert res[k][0] is None and res[k][1], k
    if res["loaded_matches"]:
        assert res["same"][0] is not None and "replayed from profiles/" in res["same"][1]
    else:  # the committed summary belongs to other kernel sources: the bench line will carry traffic: null
        import warnings
        warnings.warn("profiles/pmc_summary.json is stale for this tree's kernel sources: re-run tools/profile.sh on the GPU box")
