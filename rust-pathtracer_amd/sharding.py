"""Multi-GPU film sharding: one process per GPU, film tiles dealt along diagonals to ranks, one reduce.

The renderer itself shards inside the C ABI (`pt_render_desc.shard_index / shard_count`: rank r renders the tiles t of
the reference's tile order with PT_TILE_SHARD(t) == r — (column + row) mod N — and leaves the rest of its full-size film zero).  This module is the few lines
of `torch.distributed` plumbing around it, shared by bench.py (backend nccl = RCCL over xGMI) and by the CPU tests
(backend gloo).  Because shards are disjoint and the RNG is keyed by pixel id, the reduced film is bit-identical to the
single-process film.
"""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard(rank, world):
    """`shard=(index, count)` argument of api.render_desc for this rank ((0, 0) = whole film)."""
    return (rank, world) if world > 1 else (0, 0)


def weak_scaling_samples(spp_per_step_per_gpu, world):
    """Samples per pixel per step so that per-GPU work is independent of N: 1/N of the pixels, N x the samples."""
    return spp_per_step_per_gpu * world


def reduce_film(film, dst=0):
    """Sum the rank films into rank `dst` (the only exchange step of the whole path).  `film` is a torch tensor on the
    backend's device (HBM for nccl, host for gloo); returns it (complete on `dst` only)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():   # (also with one rank: bench.py --force-dist runs the RCCL call path on a single GPU)
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def max_over_ranks(value, device):
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value


def sum_over_ranks(values, device):
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]
