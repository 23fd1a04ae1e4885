"""tools/archive/hipcheck.py torch_first|modgpu_first -- shows that a process must use ONE HIP runtime: PyTorch
bundles its own libamdhip64.so and libmodgpu.so binds to the same SONAME, so whichever loads first
serves both (torch first: both work; libmodgpu first: torch then sees no GPU).  bench.py therefore
imports torch before the product library when it needs RCCL, and never otherwise."""
import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'libhsa' in l or 'librccl' in l})
if order == "torch_first":
    import torch
    print("torch.cuda.is_available", torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.ones(4, device="cuda"); print("torch tensor ok", x.sum().item())
    import modulate_amd as M
    print("modgpu devices", M.device_count())
    import numpy as np
    b = M.cycle_host(np.zeros(64, np.uint8), M.KEY_PS4); print("modgpu ok", b[:4])
    y = torch.ones(4, device="cuda") * 2; print("torch again", y.sum().item())
else:
    import modulate_amd as M
    import numpy as np
    print("modgpu devices", M.device_count())
    b = M.cycle_host(np.zeros(64, np.uint8), M.KEY_PS4); print("modgpu ok", b[:4])
    import torch
    print("torch.cuda.is_available", torch.cuda.is_available(), torch.cuda.device_count())
    try:
        x = torch.ones(4, device="cuda"); print("torch tensor ok", x.sum().item())
    except Exception as e:
        print("torch cuda failed:", repr(e)[:300])
print("\n".join(maps()))
