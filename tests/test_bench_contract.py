"""CPU-only: bench.py refuses to run without the HIP path (it never measures a CPU stand-in), `--gpus N` starts its own
ranks, and the loopback control plane those ranks meet on (modulate_amd/rendezvous.py) reduces and fails as it should."""
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


def test_bench_gpus_n_starts_its_own_ranks(modgpu):
    """`python bench.py --gpus 2` with no launcher environment: the parent starts two rank processes itself.  Without a GPU both
    ranks meet on the loopback rendezvous, find no HIP device and exit non-zero; the parent reports the worst status, prints no
    JSON line and does not hang.  (The same line on the GPU box: tests/test_host_gpu.py::test_bench_gpus_two_by_itself.)"""
    if modgpu.device_count() > 0:
        import pytest
        pytest.skip("GPU present")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--part-bytes", "4096"])
    assert r.returncode == 1, r.stdout + r.stderr
    assert r.stderr.count("no HIP device") == 2 and "rank 0 exited with status 1" in r.stderr and "rank 1 exited with status 1" in r.stderr
    assert '"metric"' not in r.stdout and "torch" not in r.stderr


def _plane_rank(rank, world, address, q, die_at=None):
    from modulate_amd import rendezvous, sharding
    try:
        plane = rendezvous.LoopbackPlane(rank, world, address)
        sharding.use_plane(plane)
        sharding.barrier_over_ranks()
        got = [sharding.max_over_ranks(1.0 + rank), sharding.sum_over_ranks(0.5 * (rank + 1)), sharding.max_over_ranks(-3.0 - rank)]
        if die_at == rank:
            os._exit(7)  # a rank that dies between rounds ...
        sharding.barrier_over_ranks()
        plane.close()
        q.put((rank, "ok", got))
    except Exception as e:  # ... must surface in the others as an error, not as a hang
        q.put((rank, "error", type(e).__name__))


def _run_plane(world, address, die_at=None):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_plane_rank, args=(r, world, address, q, die_at)) for r in range(world)]
    for p in ps:
        p.start()
    out = {}
    for _ in range(world - (0 if die_at is None else 1)):
        rank, status, got = q.get(timeout=60)
        out[rank] = (status, got)
    for p in ps:
        p.join(timeout=30)
        assert p.exitcode is not None
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_loopback_plane_reduces_over_ranks():
    """Both address kinds: tcp (bench.py's own launcher) and an abstract Unix socket (under torch.distributed.run)."""
    for world, address in ((1, f"tcp:127.0.0.1:{_free_port()}"), (3, f"tcp:127.0.0.1:{_free_port()}"), (8, f"unix:modgpu-test-{os.getpid()}")):
        out = _run_plane(world, address)
        assert sorted(out) == list(range(world))
        for rank, (status, got) in out.items():
            assert status == "ok", (rank, got)
            assert got == [float(world), 0.5 * world * (world + 1) / 2, -3.0], (rank, got)


def test_loopback_plane_dead_rank_is_an_error_not_a_hang():
    out = _run_plane(3, f"tcp:127.0.0.1:{_free_port()}", die_at=1)
    assert sorted(out) == [0, 2]
    assert all(status == "error" for status, _ in out.values()), out


def test_loopback_plane_default_address():
    from modulate_amd import rendezvous
    assert rendezvous.default_address({"MODGPU_BENCH_RDZV": "tcp:127.0.0.1:5"}) == "tcp:127.0.0.1:5"
    a = rendezvous.default_address({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29511", "TORCHELASTIC_RUN_ID": "x"})
    assert a == "unix:modgpu-bench-127.0.0.1-29511-x"
    import pytest
    with pytest.raises(ValueError):
        rendezvous.LoopbackPlane(0, 1, "file:/tmp/x")


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


def test_bench_under_torchrun_meets_on_the_loopback_socket(modgpu):
    """The driver's N > 1 launch line.  Under torch.distributed.run the default control plane is still the loopback socket (an
    abstract Unix socket named after MASTER_PORT, which the launcher's own store occupies): on this GPU-less box both ranks get
    through the rendezvous' first barrier and then stop at "no HIP device"."""
    if modgpu.device_count() > 0:
        import pytest
        pytest.skip("GPU present")
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MODGPU_BENCH_RDZV"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--part-bytes", "4096"], capture_output=True, text=True, env=e, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert (r.stdout + r.stderr).count("no HIP device: bench.py measures the HIP path only") == 2, r.stdout[-2000:] + r.stderr[-2000:]
    assert "nobody listens" not in r.stderr and '"metric"' not in r.stdout


def test_stopping_the_launcher_stops_its_ranks(tmp_path):
    """`python bench.py --gpus N` is a parent of N rank processes: a time limit or ^C that ends the parent must not leave ranks behind on
    the GPUs.  The ranks here are stand-ins that sleep (bench.launch_ranks starts `sys.executable bench.py ...`; the test points it at a
    sleeper through the module's own __file__), the parent gets SIGTERM, both ranks are gone within seconds and the parent exits 128 + 15."""
    import signal
    import time
    sleeper = tmp_path / "bench.py"
    sleeper.write_text("import os, sys, time\nopen(sys.argv[1] + '/pid.' + os.environ['RANK'], 'w').write(str(os.getpid()))\ntime.sleep(120)\n")
    code = ("import sys, bench\n"
            f"bench.__file__ = {str(sleeper)!r}\n"
            f"raise SystemExit(bench.launch_ranks(2, [{str(tmp_path)!r}]))\n")
    p = subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT))
    deadline = time.time() + 30
    while time.time() < deadline and not all((tmp_path / f"pid.{r}").exists() and (tmp_path / f"pid.{r}").read_text() for r in (0, 1)):
        time.sleep(0.05)
    pids = [int((tmp_path / f"pid.{r}").read_text()) for r in (0, 1)]
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    time.sleep(0.2)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = open(f"/proc/{pid}/stat").read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, pid
