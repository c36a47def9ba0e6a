"""Env sharding across ranks (SPEC.md §2, §5). Envs are independent: rank r of P owns the
contiguous global ids [r*N/P, (r+1)*N/P) and steps them with no data-path collective. Only when the
option-Q weights are shared does a step end with one all-reduce of (G, n_k) — 26 KB per VF, latency-bound."""
from __future__ import annotations

import os
from typing import Tuple

import torch


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of global env ids for `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_from_torchrun() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def allreduce_grad(G: torch.Tensor, n_k: torch.Tensor, group=None) -> None:
    """Sum (G, n_k) over ranks in place: nccl(=RCCL) for device tensors, gloo for CPU tensors."""
    import torch.distributed as dist
    dist.all_reduce(G, group=group)
    dist.all_reduce(n_k, group=group)
