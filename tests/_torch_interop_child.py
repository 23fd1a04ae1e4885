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
assert M.last_launch()["variant"] == 1, "first large launch under capture must take the static streaming shape (no ticket pair can be allocated there)"

# (4) the work-queue shape: eager on two torch streams at once (each launch gets its own ticket pair from the
# per-device ring), then captured into a graph (the pair is baked into the graph and cleans itself after every run)
n4 = (288 << 20) + 80
ta = torch.zeros(n4, dtype=torch.uint8, device="cuda")
tb = torch.zeros(n4 + 4, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for _ in range(3):  # three rounds: odd number of passes leaves the keystream in both
    M.cycle_device(ta.data_ptr(), n4, M.KEY_PS4, 0, 0, s1.cuda_stream)
    M.cycle_device(tb.data_ptr() + 4, n4, M.KEY_PS3, 12345, 0, s2.cuda_stream)
assert M.last_launch()["variant"] == 2 and M.last_launch()["kernel"].startswith("modgpu_cycle_queue_kernel<")
torch.cuda.synchronize()
for off in (0, (100 << 20) + 7, n4 - (1 << 20)):
    assert np.array_equal(ta[off:off + (1 << 20)].cpu().numpy(), O.keystream(M.KEY_PS4, 1 << 20, off)), ("stream 1", off)
    assert np.array_equal(tb[4 + off:4 + off + (1 << 20)].cpu().numpy(), O.keystream(M.KEY_PS3, 1 << 20, 12345 + off)), ("stream 2", off)
assert not tb[:4].cpu().numpy().any()
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    M.cycle_device(ta.data_ptr(), n4, M.KEY_PS4, 0, 0, torch.cuda.current_stream().cuda_stream)
assert M.last_launch()["variant"] == 2, "ticket ring exists now: the captured launch is the work-queue shape"
torch.cuda.synchronize()
for k in range(1, 4):  # ta holds the keystream: replay 1 -> zeros, 2 -> keystream, 3 -> zeros
    g3.replay()
    torch.cuda.synchronize()
    tail = ta[-(1 << 20):].cpu().numpy()
    want = np.zeros(1 << 20, np.uint8) if k % 2 else O.keystream(M.KEY_PS4, 1 << 20, n4 - (1 << 20))
    assert np.array_equal(tail, want), f"queue-shape graph replay {k}"
    if k % 2:
        assert int(ta.sum(dtype=torch.int64).item()) == 0, "every byte back to zero"
assert int(ta.sum(dtype=torch.int64).item()) == 0, "after an even number of passes in total the buffer is zero again"

# (5) several parts in one launch (modgpu_cycle_batch_device) on torch memory: eager on a side stream, then captured -- the
# part table travels in the kernel arguments, so the captured launch needs no allocation and replays with its own ticket pair
sizes5 = [(120 << 20) + 7, 5, (90 << 20) + 1, 0, (70 << 20) + 16]
ts = [torch.zeros(max(n, 1) + 8, dtype=torch.uint8, device="cuda") for n in sizes5]
ptrs5 = [t.data_ptr() + 4 for t in ts]
torch.cuda.synchronize()
stats0 = M.queue_stats()
with torch.cuda.stream(s1):
    M.cycle_batch_device(ptrs5, sizes5, M.KEY_PS3, device=0, stream=s1.cuda_stream)
assert M.last_launch()["variant"] == 3 and M.last_launch()["bytes"] == sum(sizes5)
s1.synchronize()
g5 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g5):
    M.cycle_batch_device(ptrs5, sizes5, M.KEY_PS3, device=0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
stats1 = M.queue_stats()
assert stats1["batch_launches"] - stats0["batch_launches"] == 2 and stats1["graph"] - stats0["graph"] == 1, (stats0, stats1)
for k in range(1, 4):  # eager pass left the keystream: replay 1 -> zeros, 2 -> keystream, 3 -> zeros
    g5.replay()
    torch.cuda.synchronize()
    for t, n in zip(ts, sizes5):
        got = t.cpu().numpy()
        assert not got[:4].any() and not got[4 + n:].any(), "bytes around a part untouched"
        if k % 2:
            assert not got.any(), f"batch graph replay {k}"
        elif n:
            w = min(n, 1 << 20)
            assert np.array_equal(got[4 + n - w:4 + n], O.keystream(M.KEY_PS3, w, n - w)), f"batch graph replay {k}"
            assert np.array_equal(got[4:4 + min(n, 4096)], O.keystream(M.KEY_PS3, min(n, 4096), 0))
print("TORCH_INTEROP_OK")
