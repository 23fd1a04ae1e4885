"""Child process of tests/test_gpu_torch_interop.py.  torch is imported FIRST so that its bundled
HIP runtime is the one libmodgpu.so binds to (one runtime per process)."""
import os
import sys

import torch  # noqa: E402  (must precede modulate_amd's first use)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import modulate_amd as M  # noqa: E402
from oracle import oracle as O  # noqa: E402

assert torch.cuda.is_available() and M.device_count() >= 1

# (1) torch-owned tensor, torch-owned side stream, buf+4 like the reference's callers
n = 5_000_011
pt = O.splitmix_bytes(n + 8, 77)
t = torch.from_numpy(pt.copy()).cuda()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    M.cycle_device(t.data_ptr() + 4, n, M.KEY_PS3, 0, 0, side.cuda_stream)
side.synchronize()
want = pt.copy()
O.cycle(want[4:4 + n], O.KEY_PS3)
assert np.array_equal(t.cpu().numpy(), want), "torch tensor / stream mismatch"

# (2) the launch path has no allocation or sync: capture one pass into a hipGraph and replay it
n = (3 << 20) + 123
pt = O.splitmix_bytes(n, 5)
t = torch.from_numpy(pt.copy()).cuda()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    M.cycle_device(t.data_ptr(), n, M.KEY_PS4, 0, 0, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert np.array_equal(t.cpu().numpy(), pt), "capture must record, not execute"
ct = O.cycle(pt.copy(), O.KEY_PS4)
for k in range(1, 4):
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), ct if k % 2 else pt), f"graph replay {k}"

# (3) a large buffer takes the streaming shape (barriers, pipelining) under capture as well
n = (320 << 20) + 48
t = torch.zeros(n, dtype=torch.uint8, device="cuda")
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    M.cycle_device(t.data_ptr(), n, M.KEY_PS4, 0, 0, torch.cuda.current_stream().cuda_stream)
g2.replay()
torch.cuda.synchronize()
got = t[-(1 << 20):].cpu().numpy()
assert np.array_equal(got, O.keystream(M.KEY_PS4, 1 << 20, n - (1 << 20))), "large graph replay"
print("TORCH_INTEROP_OK")
