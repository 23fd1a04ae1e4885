"""Child process of tests/test_gpu_torch_interop.py::test_every_state_in_every_byte_role.

The kernels turn one LCG state per 16-byte word into 16 keystream bytes with 16 different multipliers, and the
streaming shapes do it with the 3-instruction carry trick (cycle_kernel_impl.h ks_word_carry) whose correctness
rests on range arguments over the state.  This is the argument checked by exhaustion: the generator has
2^31 - 2 states, so ONE pass over period + 32 bytes puts every 16th state at a word's base; 16 passes whose stream
offsets differ by one byte put EVERY state there, i.e. every state meets every byte role, in each kernel shape.

Anchor: the pass at offset 0 is compared with the oracle over all of its 2 GiB.  Every other pass must equal a
shifted view of it (ks_d[i] == ks_0[i + d]) -- compared on the device, torch supplying memory and the comparison."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import torch  # noqa: E402  (first: its HIP runtime serves the process)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import modulate_amd as M  # noqa: E402
from oracle import oracle as O  # noqa: E402

M.use_testing_flavour()  # the launch shape is forced per pass
assert torch.cuda.is_available() and M.device_count() >= 1
KEY = M.KEY_PS3
P = O.PERIOD
n = P + 32
stream = torch.cuda.current_stream().cuda_stream

k0 = torch.zeros(P + 48, dtype=torch.uint8, device="cuda")
M.debug_set_launch(None)
M.cycle_device(k0.data_ptr(), k0.numel(), KEY, 0, 0, stream)
torch.cuda.synchronize()
assert M.last_launch()["variant"] == 2
host = k0.cpu().numpy()
SL = 32 << 20


def check(lo):
    ln = min(SL, host.size - lo)
    return np.array_equal(host[lo:lo + ln], O.keystream(KEY, ln, lo))


with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:  # (ctypes releases the GIL inside the oracle)
    assert all(ex.map(check, range(0, host.size, SL))), "anchor pass differs from the oracle"
assert host[P:P + 48].tobytes() == host[:48].tobytes()  # the period, seen in the anchor itself
del host
print("anchor: %d bytes == oracle" % k0.numel(), flush=True)

t = torch.empty(n, dtype=torch.uint8, device="cuda")
passes = 0
for shape, helpers in (("queue", 2), ("queue", 1), ("large", 0), ("small", 0)):
    M.debug_set_launch(shape)
    M.debug_set_helpers(helpers)
    for d in range(16):
        if shape == "queue" and helpers == 1 and d % 4:  # helper workgroups run the same code: a quarter of the shifts
            continue
        t.zero_()
        M.cycle_device(t.data_ptr(), n, KEY, d, 0, stream)
        assert M.last_launch()["variant"] == {"small": 0, "large": 1, "queue": 2}[shape], M.last_launch()
        assert torch.equal(t, k0[d:d + n]), (shape, helpers, d)
        passes += 1
    print("shape %-5s helpers=%d: every state at a word base, bytes equal the anchor's" % (shape, helpers), flush=True)
# one more key with the sign bit set (the reference's int arithmetic on a negative key), against the oracle on windows
M.debug_set_launch(None)
M.debug_set_helpers(0)
for key in (M.KEY_PS4, 0x80000001):
    t.zero_()
    M.cycle_device(t.data_ptr(), n, key, 5, 0, stream)
    torch.cuda.synchronize()
    for off in (0, (1 << 30) + 3, n - (1 << 20)):
        assert np.array_equal(t[off:off + (1 << 20)].cpu().numpy(), O.keystream(key, 1 << 20, 5 + off)), (hex(key), off)
print("EVERY_STATE_OK passes=%d" % passes)
