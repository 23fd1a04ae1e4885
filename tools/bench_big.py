import os, sys
sys.path.insert(0, os.getcwd())
os.environ["MODGPU_REQUIRE_GPU"] = "1"
import modulate_amd as M
print("one call over a single resident buffer (modgpu_time_cycle_device, 2 launches, HIP events); TB/s = 2 x bytes / time")
for gib in (1, 4, 8, 16, 48, 96, 160, 224):
    n = gib << 30
    try:
        b = M.DeviceBuffer(n)
    except M.ModGpuError as e:
        print("%4d GiB: allocation failed (%s)" % (gib, str(e)[:60])); break
    M.time_cycle_device(b.ptr, n, M.KEY_PS4, 0, 0, None, iters=2)
    ms = M.time_cycle_device(b.ptr, n, M.KEY_PS4, 0, 0, None, iters=2)
    info = M.last_launch()
    print("%4d GiB  %9.3f ms per launch  %6.3f TB/s   grid %d (%d main)  %s" % (gib, ms, 2.0 * n / (ms * 1e-3) / 1e12, info["grid"], info["main_groups"], info["kernel"].split("<")[0]), flush=True)
    b.free()
