"""Multi-GPU sharding: one process per GPU, envs split into contiguous shards, no per-step traffic.

Episodes are independent, so the only exchange is one all-gather of the per-env episode
returns / mean Strehl at episode end (RCCL over xGMI via torch.distributed backend "nccl"; "gloo" on CPU
for the tests)."""
from __future__ import annotations

import os


def shard_bounds(n_envs_total: int, rank: int, world_size: int):
    """Contiguous shard [lo, hi) of rank; the first ``n % world`` ranks hold one extra env."""
    if not (0 <= rank < world_size):
        raise ValueError("rank outside [0, world_size)")
    base, extra = divmod(int(n_envs_total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_from_env(backend: str | None = None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def all_gather_returns(local, n_envs_total: int):
    """Gathers the per-env vector ``local`` (shape [n_local]) of every rank into the global vector
    [n_envs_total], ordered by global env index.  Shards may differ by one env: pad, gather, trim."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    width = -(-n_envs_total // world)
    buf = torch.zeros(width, dtype=local.dtype, device=local.device)
    buf[:local.numel()] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_envs_total, r, world)
        parts.append(out[r][:hi - lo])
    return torch.cat(parts)
