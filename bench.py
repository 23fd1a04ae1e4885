#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config 2/3 workload.

A "step" is one encrypt pass + one decrypt pass of the cipher hot path
(CEncryptionCycler::Cycle, Modulate/CEncryptionCycler.cpp:4-25) over one synthetic 4 GiB
(2^32-byte) .ark part that is already resident in HBM: two kernel launches through the C ABI
(modgpu_cycle_device).  With N GPUs every rank owns one such part (parts are independent
streams: no collective on the data path, weak scaling).

    python bench.py --gpus N --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The first form needs no launcher: with N > 1 and no WORLD_SIZE in the environment this process never touches the GPU -- it
starts N fresh rank processes (one per GPU), relays rank 0's line and exits with the worst of their statuses.  Either way the
ranks meet for the contract's barrier and MAX over a loopback socket (modulate_amd/rendezvous.py): no torch, no RCCL -- the
workload has no exchange step.  `--backend gloo|nccl` puts torch.distributed there instead.

Rank 0 prints ONE JSON line.  `value` = part bytes cycled per second summed over all ranks
(each pass over a byte counts once; HBM traffic is twice that: 1 read + 1 write per byte).
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Read once when libmodgpu.so is loaded: forbids the library's host loop, so that nothing this script
# times or checks can have been computed anywhere but on the GPU (the run fails instead).
os.environ["MODGPU_REQUIRE_GPU"] = "1"
# --backend nccl only: the host driver of this pool supports dmabuf IPC only (RCCL's barrier shares device memory across ranks)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample_bytes):
    """The reference's single-thread loop on this node's host cores, timed on a bounded sample
    (encrypt + decrypt of `sample_bytes`), the thread pinned to one core.  Uses the compiled
    reference (oracle/_ref) when its prebuilt library travelled with the repo, else our C restatement
    of it (the oracle).  This is the only place bench.py touches oracle/ -- as the thing timed beside
    the GPU, never as the thing measured in `value`."""
    from oracle import oracle as O
    buf = O.splitmix_bytes(sample_bytes, 7)
    kind = "reference" if O.have_ref() else "port"
    fn = O.ref_cycle if kind == "reference" else O.cycle
    pinned_to = None
    old_affinity = None
    try:
        old_affinity = os.sched_getaffinity(0)
        pinned_to = max(old_affinity)  # away from core 0, which takes most interrupts
        os.sched_setaffinity(0, {pinned_to})
    except (AttributeError, OSError):
        pinned_to = None
    try:
        fn(buf[:1 << 16].copy(), O.KEY_PS4)  # load + warm
        t0 = time.perf_counter()
        fn(buf, O.KEY_PS4)
        fn(buf, O.KEY_PS4)
        dt = time.perf_counter() - t0
    finally:
        if old_affinity is not None:
            try:
                os.sched_setaffinity(0, old_affinity)
            except OSError:
                pass
    out = {"value": round(2 * sample_bytes / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": kind,
           "cpu_model": cpu_model(), "pinned_to_core": pinned_to, "host_logical_cores": os.cpu_count(),
           "sample": f"{sample_bytes >> 20} MiB part, encrypt+decrypt (2 passes), 1 thread"
                     f"{'' if pinned_to is None else f' pinned to core {pinned_to}'}, {dt:.1f} s"}
    # informational second line ("all host cores, one part per core"): the same loop on the cores this process
    # may use (at most 64 threads), one independent 64 MiB part per thread (the checker's C call releases the GIL)
    try:
        from concurrent.futures import ThreadPoolExecutor
        threads = min(64, len(old_affinity) if old_affinity else (os.cpu_count() or 1))  # (bounded: a few seconds of CPU work)
        per = 64 << 20
        parts = [buf[:per].copy() for _ in range(threads)]
        with ThreadPoolExecutor(max_workers=threads) as ex:
            t0 = time.perf_counter()
            list(ex.map(lambda p: (fn(p, O.KEY_PS4), fn(p, O.KEY_PS4)), parts))
            dta = time.perf_counter() - t0
        out["all_cores"] = {"value": round(2 * per * threads / dta / 1e9, 2), "unit": "GB/s", "cores": threads,
                            "sample": f"{threads} x 64 MiB parts, encrypt+decrypt, one part per thread, {dta:.1f} s"}
    except Exception as e:  # informational only: never fails the bench
        out["all_cores"] = {"value": None, "error": str(e)[:120]}
    return out


def load_traffic(n_bytes, kernel, source_hash):
    """HBM bytes per launch from the committed PMC summary (profiles/pmc_summary.json) -- but only if that
    summary was taken on THIS device code (same kernel-source hash, same instantiation, same part size).
    Counter passes cannot run inside a timed bench, so `traffic` is a replayed figure; `traffic_source`
    says which profile it is, and anything that does not match the loaded library yields null.
    Returns (traffic, traffic_source)."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        with open(path) as f:
            s = json.load(f)
    except (OSError, ValueError):
        return None, "no committed PMC summary"
    tag = s.get("tag", "?")
    if int(s.get("part_bytes", -1)) != n_bytes:
        return None, f"profiles/{tag} PMC is for part_bytes={s.get('part_bytes')}, not measured for this size"
    if not source_hash or s.get("kernel_source_hash") != source_hash:
        return None, f"profiles/{tag} PMC was taken on other kernel sources ({str(s.get('kernel_source_hash'))[:12]}... vs loaded {str(source_hash)[:12]}...)"
    if kernel not in str(s.get("cycle_kernel", "")):
        return None, f"profiles/{tag} PMC was taken on {s.get('cycle_kernel')}"
    return s.get("hbm_bytes_per_launch"), (f"replayed from profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                           f"separate passes, same kernel-source hash {source_hash[:12]}...)")


def load_valu(n_bytes, kernel, source_hash):
    """VALU utilisation of the same launches ("report VALU utilisation beside HBM %"): the `valu` object of the
    committed PMC summary, under the same conditions as `traffic` (null otherwise)."""
    if load_traffic(n_bytes, kernel, source_hash)[0] is None:
        return None
    with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as f:
        return json.load(f).get("valu")


class _StdoutToStderr:
    """While active, file descriptor 1 points at stderr.  RCCL prints a version banner ("RCCL version : ...", five lines) to
    the process's stdout from C when a communicator is set up; the bench contract wants ONE JSON line there (VERDICT r3 weak
    #9a).  The process group's set-up and its first collective run inside this; the banner lands on stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` by itself: start N NEW processes of this script, rank r on GPU r, and wait for them.
    This process stays off the GPU (it never imports modulate_amd); no process that has opened the GPU is replaced or
    re-executed; a failing rank is a non-zero exit, upon which the others are ended by PID.  Rank 0's stdout is relayed as the
    last thing on this process's stdout; the other ranks' stdout goes to stderr.  Returns the worst child status."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MODGPU_BENCH_RDZV=f"tcp:127.0.0.1:{port}", MODGPU_BENCH_LAUNCHER="bench.py")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    def stop_ranks(signum, _frame):  # whoever stops this process (a time limit, ^C) stops the ranks it started -- by their PIDs
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10.0)
            except subprocess.TimeoutExpired:
                p.kill()
        os._exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, stop_ranks)
    rank0_out = []
    reader = threading.Thread(target=lambda: rank0_out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    worst = 0
    live = set(range(n_ranks))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            rc = 128 - rc if rc < 0 else rc  # ended by a signal
            if rc != 0:
                print(f"bench.py: rank {r} exited with status {rc}", file=sys.stderr)
                worst = max(worst, rc)
        if worst and live:  # one rank failed: the others would wait for it at the next barrier
            for r in live:
                procs[r].terminate()
            deadline = time.monotonic() + 10.0
            for r in live:
                try:
                    procs[r].wait(timeout=max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        time.sleep(0.05)
    reader.join(timeout=10.0)
    sys.stdout.write("".join(rank0_out))
    sys.stdout.flush()
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--part-bytes", type=int, default=1 << 32, help="bytes per .ark part (default 4 GiB)")
    ap.add_argument("--key", type=lambda s: int(s, 0), default=0x90CFC0AB)
    ap.add_argument("--cpu-sample-bytes", type=int, default=3 << 29, help="bytes of the bounded CPU-baseline sample (default 1.5 GiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-first-pass", action="store_true", help="skip the first-pass preamble (14 launches before the warm-up)")
    ap.add_argument("--no-own-upload-probe", action="store_true", help="skip the fresh child process that uploads with its own hipMemcpy and calls "
                    "modgpu_prepare (first_pass.callers_own_upload_after_modgpu_prepare; N = 1 only; tools/profile.sh skips it)")
    ap.add_argument("--backend", default="socket", choices=("socket", "gloo", "nccl"),
                    help="control plane for the barrier / MAX over ranks: socket = a loopback rendezvous, no torch (default); "
                         "gloo = torch.distributed on CPU tensors; nccl = RCCL on device tensors")
    ap.add_argument("--force-device", type=int, default=None, help="rehearsal only: every rank uses this HIP device")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal only: bring the control plane up even with one rank")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    if a.force_dist:
        os.environ["MODGPU_BENCH_FORCE_DIST"] = "1"
    use_control_plane = int(os.environ.get("WORLD_SIZE", "1")) > 1 or a.force_dist
    use_torch = use_control_plane and a.backend != "socket"

    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so, libmodgpu.so binds to the
    # same SONAME.  Whichever is loaded first serves both, so when torch.distributed is the control plane torch
    # is imported BEFORE the product library is first used; otherwise torch is never imported.
    if use_torch:
        import torch  # noqa: F401
    import modulate_amd as M
    from modulate_amd import sharding

    rank, local_rank, world = sharding.dist_env()
    a.gpus = world
    dist = None
    plane = None
    red_dev = None
    launcher = os.environ.get("MODGPU_BENCH_LAUNCHER") or ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "none")
    control_plane = "none (one rank)"
    if use_control_plane and not use_torch:
        from modulate_amd import rendezvous
        plane = rendezvous.LoopbackPlane(rank, world)
        sharding.use_plane(plane)
        plane.barrier()
        control_plane = f"socket: loopback rendezvous ({plane.address.split(':')[0]}), standard library only -- no torch, no RCCL"
    elif use_torch:
        import torch
        import torch.distributed as dist
        control_plane = "nccl: torch.distributed over RCCL, device tensors" if a.backend == "nccl" else "gloo: torch.distributed, CPU tensors"
        with _StdoutToStderr():
            if a.backend == "nccl":
                # one rank per GPU; if the launcher already narrowed each rank's visibility to one device
                # (HIP_VISIBLE_DEVICES per rank), that device is index 0 for everybody
                if torch.cuda.device_count() <= local_rank:
                    local_rank = 0
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                red_dev = torch.device("cuda", local_rank)
            else:
                dist.init_process_group(a.backend)
            dist.barrier()  # the first collective sets the communicator up (and prints RCCL's banner): here, off stdout
    if M.device_count() < 1:
        raise SystemExit("no HIP device: bench.py measures the HIP path only")
    dev = local_rank if a.force_device is None else a.force_device
    if dev >= M.device_count():
        raise SystemExit(f"rank {rank}: HIP device {dev} does not exist ({M.device_count()} visible): --gpus N wants N GPUs on this node "
                         f"(--force-device D rehearses N ranks on one)")

    # A caller that brings its OWN device memory (hipMalloc / hipMemcpy of its own) gets the library's device preparation and wake-up
    # by name: modgpu_prepare (ABI 8).  Measured in a fresh child process -- a process's first launch is the point -- before this
    # process has queued anything, N = 1 only: tools/first_launch_own_upload.py.  Outside every timed region; reporting only.
    own_upload = None
    if world == 1 and not a.no_first_pass and not a.no_own_upload_probe and a.force_device in (None, 0):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "first_launch_own_upload.py"), "--prepare", "--bytes", str(min(a.part_bytes, 411 * 1000 * 1000))],
                               capture_output=True, text=True, timeout=120, cwd=ROOT)
            own_upload = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-200:]}
        except Exception as e:  # informational: never fails the bench
            own_upload = {"error": str(e)[:200]}

    n = a.part_bytes
    # synthetic part: uniform random bytes (values do not affect timing; they make the checks real)
    rng = np.random.default_rng(0x4D6F6475 + rank)
    tile = rng.integers(0, 256, size=min(n, 64 << 20), dtype=np.uint8)
    part = M.DeviceBuffer(n, device=dev)
    for off in range(0, n, tile.size):
        part.upload(tile[:min(tile.size, n - off)], offset=off)

    def barrier():
        part.sync()
        sharding.barrier_over_ranks()
        part.sync()

    # ---- the first-pass regime (outside the timed region; reported beside the steady-state figure).
    # A real job makes ONE pass per part right after something else wrote it (BASELINE configs 3-5); the timed region
    # below is a 2K-launch steady state.  Here: this fresh process has just uploaded the part; one launch over a
    # 411 MB slice (config 4's part size), undone by a second; then twelve single launches over the whole part, each
    # timed by its own pair of HIP events on the launch stream.  Launch 1 is "the first pass"; launches 3..10 show the
    # chip's clock transient (profiles/r03_first_pass.txt: the shader clock drops from its idle boost to ~1.45 GHz about
    # 2 ms into the load and climbs back over ~15 ms; this kernel follows the clock below ~1.9 GHz).  An even number of
    # launches, so the part is plaintext again afterwards.
    first_pass = None
    launches_before_timed = 0
    if not a.no_first_pass:
        small_n = min(n, 411 * 1000 * 1000)
        part.sync()
        t_small = M.time_cycle_device(part.ptr, small_n, a.key, 0, dev, None, iters=1)
        t_small_again = M.time_cycle_device(part.ptr, small_n, a.key, 0, dev, None, iters=1)
        time.sleep(0.05)  # idle again, as after an upload
        series = [M.time_cycle_device(part.ptr, n, a.key, 0, dev, None, iters=1) for _ in range(12)]
        launches_before_timed += 14

        def rate(nbytes, ms):
            return {"bytes": nbytes, "ms": round(ms, 4), "achieved": round(2.0 * nbytes / (ms * 1e-3) / 1e9, 1),
                    "frac": round(2.0 * nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        first_pass = {"what": "fresh process, part just uploaded: ONE encrypt launch, HIP events on the launch stream (GB/s = 2*bytes/time)",
                      "part": rate(n, series[0]),
                      # one launch from an IDLE queue: the event pair also holds what lies between the start marker and the kernel's
                      # first wave (the host's planning and packet write, the chip bringing its shader engines up) -- the dispatch itself
                      # takes 0.127-0.128 ms here by rocprofv3's timestamps, 0.80-0.81 of peak (profiles/r05_first_launch.txt, run F)
                      "part_411MB": rate(small_n, t_small), "part_411MB_next_launch": rate(small_n, t_small_again),
                      # the same first launch for the two kinds of caller (include/modgpu.h, device-memory helpers): one that uploads through
                      # modgpu_alloc / modgpu_h2d (this process: == part_411MB), and one with its own hipMalloc / hipMemcpy that calls
                      # modgpu_prepare when its upload starts (a fresh child process, tools/first_launch_own_upload.py)
                      "via_library_upload": rate(small_n, t_small),
                      "callers_own_upload_after_modgpu_prepare": own_upload,
                      "part_411MB_note": "events from an idle queue include dispatch latency; the dispatch itself: 0.127-0.128 ms = 0.80-0.81 (profiles/r05_first_launch.txt run F)",
                      "ms_of_launches_1_to_12": [round(x, 4) for x in series],
                      "slowest_of_launches_1_to_12": rate(n, max(series)),
                      "cause_of_the_dip": "shader-clock (DVFS) transient after load onset, not the buffer's state: profiles/r03_first_pass.txt"}

    # The first collective of a process group sets the communicator up (seconds, GPU idle).  Do that here, not in the
    # barrier in front of the timed region: the chip would come to the timed launches from idleness, i.e. straight into
    # its clock transient (profiles/r03_first_pass.txt), which the warm-up exists to get out of the way.
    barrier()
    # warmup
    for _ in range(a.warmup):
        part.cycle(a.key)
        part.cycle(a.key)
    launches_before_timed += 2 * a.warmup
    barrier()
    # timed region: exactly `steps` steps = 2*steps launches, HIP events on the launch stream
    t0 = time.perf_counter()
    ms_per_launch = M.time_cycle_device(part.ptr, n, a.key, 0, dev, None, iters=2 * a.steps)
    t_sync = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get("MODGPU_BENCH_DEBUG"):
        print(f"[rank {rank}] steps done+synced at {t_sync*1e3:.3f} ms, after trailing barrier {dt*1e3:.3f} ms", file=sys.stderr)
    dt = sharding.max_over_ranks(dt, red_dev)
    ms_per_launch = sharding.max_over_ranks(ms_per_launch, red_dev)
    launch = M.last_launch()  # what the library launched for this size, straight from its planner

    # post-run checks (outside the timed region; no oracle code here -- committed golden DATA only):
    # an even number of passes must give the original bytes back (involution), and after one more
    # pass  ciphertext ^ plaintext  must equal the reference's own keystream samples
    # (tests/golden/cycle_golden.json, generated from the compiled reference; PS4 key, offsets < n).
    ok = True
    for off in (0, max(0, n // 2 - 4096), max(0, n - (1 << 20))):
        ln = min(1 << 20, n - off)
        ok = ok and bool(np.array_equal(part.download(ln, offset=off), np.resize(np.roll(tile, -(off % tile.size)), ln)))
    part.cycle(a.key)
    part.sync()
    checked = 0
    if a.key == 0x90CFC0AB:
        with open(os.path.join(ROOT, "tests", "golden", "cycle_golden.json")) as f:
            gold = json.load(f)
        samples = [{"off": 0, "hex": gold["keystream"][1]["first64"]}] + list(gold["large"]["samples"]) + \
                  [{"off": gold["large"]["around_period"]["start"], "hex": gold["large"]["around_period"]["hex"]},
                   {"off": gold["large"]["tail16"]["start"], "hex": gold["large"]["tail16"]["hex"]}]
        assert gold["keystream"][1]["key"] == 0x90CFC0AB
        for smp in samples:
            m = len(smp["hex"]) // 2
            if smp["off"] + m > n:
                continue
            pt = np.resize(np.roll(tile, -(smp["off"] % tile.size)), m)
            ks = part.download(m, offset=smp["off"]) ^ pt
            ok = ok and ks.tobytes().hex() == smp["hex"]
            checked += 1
    part.cycle(a.key)
    part.sync()
    n_ok = sharding.sum_over_ranks(1.0 if ok else 0.0, red_dev)

    if rank == 0:
        total_bytes = float(world) * a.steps * 2 * n
        achieved = 2.0 * n / (ms_per_launch * 1e-3) / 1e9  # read + write per launch
        traffic, traffic_source = load_traffic(n, launch["kernel"], M.kernel_source_hash())
        stats = M.path_stats()
        out = {
            "metric": "GB/s encrypt+decrypt over synthetic .ark parts; % HBM peak at 1/2/4/8 GPU",
            "value": round(total_bytes / dt / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"config {'2' if world == 1 else '3'}: {world} x {n} B synthetic .ark part"
                                   f"{'s, one per GPU' if world > 1 else ''}, encrypt pass + decrypt pass per step, "
                                   f"HBM-resident, key {a.key:#010x}",
                       "part_bytes": n, "passes_per_step": 2, "parallelism": f"parts{world}",
                       "control_plane": control_plane, "launcher": launcher,
                       "value_counts": "payload bytes cycled per second (HBM read+write traffic is 2x)",
                       "bit_exact_check": ("pass" if n_ok == world else "FAIL") + f" (involution + {checked} golden keystream samples per rank)",
                       "engine": f"gfx950 kernel only (MODGPU_REQUIRE_GPU=1): {stats['gpu_launches']} launches, "
                                 f"{stats['scalar_calls']} host-loop calls on rank 0"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "valu": load_valu(n, launch["kernel"], M.kernel_source_hash()),
                         "kernel": launch["kernel"], "grid": launch["grid"], "main_workgroups": launch["main_groups"], "block": launch["block"],
                         "chunk_bytes": launch["chunk_bytes"], "kernel_source_hash": M.kernel_source_hash(),
                         "ms_per_launch": round(ms_per_launch, 4), "algorithmic_bytes_per_launch": 2 * n,
                         # which launches of this kernel (in launch order, from 0) the HIP events of the timed region bracket:
                         # tools/summarize_profile.py averages rocprofv3's traced durations over exactly these
                         "timed_launches": [launches_before_timed, launches_before_timed + 2 * a.steps]},
        }
        if first_pass is not None:
            out["roofline"]["first_pass"] = first_pass
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample_bytes)
    if plane is not None:
        plane.barrier()
        plane.close()
    if dist is not None:
        with _StdoutToStderr():  # (nothing of the teardown may reach stdout either)
            dist.barrier()
            dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)  # the last thing this process writes to stdout, and the only one
    part.free()
    if n_ok != world:
        raise SystemExit("bit-exact check FAILED")


if __name__ == "__main__":
    main()
