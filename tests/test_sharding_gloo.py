"""CPU, world_size 2 and 8 over gloo: the multi-GPU path is a partition with no data-path collective.
The ranks here stand in for GPUs: each cycles the parts it owns with the PRODUCT's host loop
(modgpu_cycle_scalar_host -- there is no GPU in this container) and the parent checks every part
against the oracle.  Under test: round-robin part ownership, stream splitting by offset with
stream_off, and the MAX / SUM reductions of the bench contract -- the host logic bench.py and
modgpu_cycle_parts_host share.  (The N-worker GPU code itself runs in tests/test_gpu_parity.py::
test_eight_workers_on_aliased_devices on the GPU box.)"""
import os
import socket

import numpy as np
import pytest

from modulate_amd import sharding


def test_parts_for_rank_partition():
    for n_parts in (0, 1, 2, 7, 8, 9, 100):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                mine = sharding.parts_for_rank(n_parts, r, world)
                assert all(i % world == r for i in mine)
                seen += mine
            assert sorted(seen) == list(range(n_parts))
    with pytest.raises(ValueError):
        sharding.parts_for_rank(4, 2, 2)


def test_split_stream_covers():
    for n in (0, 1, 15, 16, 17, 4096, 1_000_003, 1 << 32):
        for world in (1, 2, 3, 8):
            spans = sharding.split_stream(n, world)
            assert len(spans) == world and sum(l for _, l in spans) == n
            pos = 0
            for off, ln in spans:
                assert off == min(pos, n) or ln == 0
                assert off % 16 == 0 or ln == 0
                pos = off + ln


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    import torch.distributed as dist
    import modulate_amd as M
    from oracle import oracle as O  # input generator only in the workers; the parent does the checking
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert sharding.dist_env() == (rank, rank, world)
        # (1) independent parts, part i -> rank i mod N, no exchange
        sizes = [5000, 0, 70001, 4096, 33]
        for i in sharding.parts_for_rank(len(sizes), rank, world):
            part = O.splitmix_bytes(sizes[i], 100 + i)
            M.cycle_scalar_host(part, M.KEY_PS4)  # each part is its own Cycle from offset 0
            np.save(os.path.join(tmpdir, f"part{i}.npy"), part)
        # (2) one stream split across ranks by byte offset
        n = 200_003
        off, ln = sharding.split_stream(n, world)[rank]
        seg = O.splitmix_bytes(n, 9)[off:off + ln].copy()
        M.cycle_scalar_host(seg, M.KEY_PS3, stream_off=off)
        np.save(os.path.join(tmpdir, f"seg{rank}.npy"), seg)
        # (3) bench contract reductions
        assert sharding.max_over_ranks(1.0 + rank) == float(world)
        assert sharding.sum_over_ranks(1.0) == float(world)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(os.environ.get("MODGPU_REQUIRE_GPU", "0") not in ("", "0"), reason="MODGPU_REQUIRE_GPU forbids the host loop the CPU ranks use")
@pytest.mark.parametrize("world", [2, 8])
def test_world_n_gloo(tmp_path, oracle, world):
    """world 2, and world 8 -- the node's GPU count, with more ranks than parts (5): three ranks own nothing and must still take
    part in the reductions and the barrier; the split stream has eight spans."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sizes = [5000, 0, 70001, 4096, 33]
    for i, s in enumerate(sizes):
        got = np.load(tmp_path / f"part{i}.npy")
        assert np.array_equal(got, oracle.cycle(oracle.splitmix_bytes(s, 100 + i), oracle.KEY_PS4))
    n = 200_003
    whole = np.concatenate([np.load(tmp_path / f"seg{r}.npy") for r in range(world)])
    assert np.array_equal(whole, oracle.cycle(oracle.splitmix_bytes(n, 9), oracle.KEY_PS3))
