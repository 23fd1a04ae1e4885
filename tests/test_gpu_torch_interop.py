"""GPU: the C ABI on memory and streams owned by PyTorch (plumbing only: torch supplies the device
buffer and the stream, the arithmetic is the HIP kernel), and hipGraph capture of the launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu(modgpu):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available() and modgpu.device_count() >= 1
    return torch


def test_torch_tensor_on_torch_stream(torch_gpu, modgpu, oracle):
    torch = torch_gpu
    n = 5_000_011
    pt = oracle.splitmix_bytes(n + 8, 77)
    t = torch.from_numpy(pt.copy()).cuda()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        modgpu.cycle_device(t.data_ptr() + 4, n, modgpu.KEY_PS3, 0, 0, side.cuda_stream)  # buf+4, like the reference's callers
    side.synchronize()
    want = pt.copy()
    oracle.cycle(want[4:4 + n], oracle.KEY_PS3)
    assert np.array_equal(t.cpu().numpy(), want)


def test_launch_is_graph_capturable(torch_gpu, modgpu, oracle):
    """No allocation / sync inside the launch path: one pass captured into a hipGraph, replayed."""
    torch = torch_gpu
    n = (3 << 20) + 123
    pt = oracle.splitmix_bytes(n, 5)
    t = torch.from_numpy(pt.copy()).cuda()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        modgpu.cycle_device(t.data_ptr(), n, modgpu.KEY_PS4, 0, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), pt)  # capture records, it does not execute
    ct = oracle.cycle(pt.copy(), oracle.KEY_PS4)
    for k in range(1, 4):
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), ct if k % 2 else pt), k  # involution: odd replays = ciphertext
