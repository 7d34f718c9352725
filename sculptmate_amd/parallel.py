"""Data-parallel plumbing: one process per GPU, images sharded over ranks, no data-path collective.

The path shards over images: image i -> rank i mod world; every rank holds a full
weight replica.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in CPU tests) is used only to
agree on the wall time of a step batch (barrier + MAX) and to gather per-rank mesh counts.
"""
import os

import torch


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, device=None):
    """Initialise the default process group from the torchrun environment (MASTER_ADDR defaults to 127.0.0.1)."""
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return dist


def shard_indices(n_items, rank, world):
    """Indices of the items rank `rank` processes: i with i mod world == rank (round robin)."""
    return list(range(rank, n_items, world))


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX of a Python float over all ranks (the timing contract of bench.py)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(counts, device="cpu"):
    """All ranks' integer tuples (e.g. (n_verts, n_faces) per mesh) -> list indexed by rank."""
    import torch.distributed as dist

    t = torch.tensor(list(counts), dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return [t.tolist()]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]
