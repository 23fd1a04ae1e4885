import os, sys, json, subprocess
ROOT = os.getcwd()
code_torch_first = "import torch, runpy, sys; sys.argv=['bench.py','--no-cpu-baseline','--no-first-pass']; runpy.run_path('bench.py', run_name='__main__')"
def run(label, env_extra, torch_first):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-c", code_torch_first] if torch_first else [sys.executable, "bench.py", "--no-cpu-baseline", "--no-first-pass"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT)
    try:
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]); ro = d["roofline"]
        print("%-46s value %8.2f  ms/launch %.4f  step-2*launch %.1f us" % (label, d["value"], ro["ms_per_launch"], (d["ms_per_step"] - 2 * ro["ms_per_launch"]) * 1e3), flush=True)
    except Exception as e:
        print(label, "FAILED", r.stderr[-300:])
for rep in range(2):
    run("system runtime (plain)", {}, False)
    run("system runtime, HIP_FORCE_DEV_KERNARG=0", {"HIP_FORCE_DEV_KERNARG": "0"}, False)
    run("system runtime, HIP_FORCE_DEV_KERNARG=1", {"HIP_FORCE_DEV_KERNARG": "1"}, False)
    run("torch's runtime (torch imported first)", {}, True)
    run("torch's runtime, HIP_FORCE_DEV_KERNARG=1", {"HIP_FORCE_DEV_KERNARG": "1"}, True)
    run("torch's runtime, HIP_FORCE_DEV_KERNARG=0", {"HIP_FORCE_DEV_KERNARG": "0"}, True)
