"""Env sharding across ranks (SPEC.md §2, §5) and the few collectives the sharded agent needs.

Envs are independent: rank r of P owns the contiguous global ids [r*N/P, (r+1)*N/P) and steps them with no
data-path collective. Only when the option-Q weights are shared does a step end with ONE all-reduce of the packed
operand (G followed by the update counts as floats, 26 KB per VF: latency-bound). The outer skill-discovery loop of
a sharded agent takes its decisions on all-reduced counts and fits every initiation set on the examples of all
ranks, so that every rank issues the same collectives and holds the same classifier table."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of global env ids for `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _via_host(t: torch.Tensor, group) -> bool:
    """gloo carries CPU tensors only: device tensors are staged through the host (CPU rehearsal of the N>1 path)."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


def allreduce_packed(gp: torch.Tensor, group=None) -> None:
    """Sum the packed gradient operand [n_vf*5*1296 + n_vf] (G, then the update counts as floats — exact, they stay
    far below 2^24) over the ranks of `group`, in place: nccl (= RCCL over xGMI) on device tensors, gloo via the host."""
    import torch.distributed as dist
    if _via_host(gp, group):
        gc = gp.cpu()
        dist.all_reduce(gc, group=group)
        gp.copy_(gc)
    else:
        dist.all_reduce(gp, group=group)


def allgather_packed(gp: torch.Tensor, slots: torch.Tensor, group=None) -> None:
    """Every rank's packed operand into `slots` [world, len(gp)], row r = rank r — the operand of the order-pinned sum
    (scg_apply_update_slots): ONE all-gather per step instead of the all-reduce; weights identical on every rank of a run and
    reproducible by the oracle for any number of ranks (the rank count is part of the run's identity)."""
    import torch.distributed as dist
    if _via_host(gp, group):
        sc = torch.empty(slots.numel(), dtype=slots.dtype)
        dist.all_gather_into_tensor(sc, gp.cpu(), group=group)
        slots.copy_(sc.view(slots.shape))
    else:
        dist.all_gather_into_tensor(slots.view(-1), gp, group=group)      # (the concatenated form: row r = rank r)


def allreduce_sum_int(value: int, group, device) -> int:
    """Sum of a host integer over the ranks (loop decisions of the sharded outer loop must agree on every rank)."""
    import torch.distributed as dist
    gloo = dist.get_backend(group) == "gloo"
    t = torch.tensor([int(value)], dtype=torch.int64, device="cpu" if gloo else device)
    dist.all_reduce(t, group=group)
    return int(t.item())


def allreduce_max_int(value: int, group, device) -> int:
    """Maximum of a host integer over the ranks."""
    import torch.distributed as dist
    gloo = dist.get_backend(group) == "gloo"
    t = torch.tensor([int(value)], dtype=torch.int64, device="cpu" if gloo else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def allgather_rows(t: torch.Tensor, group) -> torch.Tensor:
    """Concatenate the rows of `t` ([n_r, ...], n_r differing per rank) over the ranks, in rank order."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    gloo = dist.get_backend(group) == "gloo"
    src = t.cpu() if (gloo and t.is_cuda) else t
    n = torch.tensor([src.shape[0]], dtype=torch.int64, device=src.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    pad = torch.zeros((m,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[: src.shape[0]] = src
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.cat([p[:s] for p, s in zip(parts, sizes)])
    return out.to(t.device)
