"""Part -> GPU assignment and the cross-rank timing reduction used by bench.py.

Archive parts are independent cipher streams (each is its own Cycle call from stream offset 0;
the reference reads and writes them as separate files, Modulate/CArk.cpp:741-755, 849-897), so
the multi-GPU path is a pure partition: part i goes to rank i mod N, nothing is exchanged on the
data path.  The only cross-rank traffic is the bench's barrier and the MAX over per-rank times, which go over a
loopback socket (rendezvous.py) by default and over torch.distributed (gloo / RCCL) on request.
"""
import os


def parts_for_rank(n_parts, rank, world):
    """Indices of the parts rank `rank` of `world` owns (round-robin, as modgpu_cycle_parts_host)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_parts, world))


def split_stream(n_bytes, world, align=16):
    """Cut ONE logical stream of n_bytes into `world` contiguous [offset, length) spans (used when
    there are fewer parts than GPUs: every span is cycled with stream_off = its offset)."""
    per = -(-n_bytes // world)
    per = -(-per // align) * align
    spans = []
    for r in range(world):
        off = min(r * per, n_bytes)
        spans.append((off, min(per, n_bytes - off)))
    return spans


def dist_env():
    """(rank, local_rank, world) from the torch.distributed.run environment (defaults 0,0,1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


_plane = None  # the control plane bench.py installed (rendezvous.LoopbackPlane); None = torch.distributed, if initialised


def use_plane(plane):
    """Route barrier / MAX / SUM through `plane` (an object with barrier(), max(v), sum(v)); None goes back to torch.distributed."""
    global _plane
    _plane = plane


def _single_process():
    # torch is only imported when there really are several ranks: importing it after libmodgpu.so
    # has loaded the system HIP runtime would bring torch's bundled copy in beside it (see bench.py)
    return int(os.environ.get("WORLD_SIZE", "1")) <= 1 and not os.environ.get("MODGPU_BENCH_FORCE_DIST")


def _torch_reduce(value, device, op_name):
    if _single_process():
        return float(value)
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op_name))
    return float(t.item())


def max_over_ranks(value, device=None):
    """MAX-reduce a python float over the ranks (identity with one rank and no control plane)."""
    if _plane is not None:
        return float(_plane.max(value))
    return _torch_reduce(value, device, "MAX")


def sum_over_ranks(value, device=None):
    if _plane is not None:
        return float(_plane.sum(value))
    return _torch_reduce(value, device, "SUM")


def barrier_over_ranks():
    """The bench contract's barrier (no-op with one rank and no control plane)."""
    if _plane is not None:
        _plane.barrier()
        return
    if _single_process():
        return
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
