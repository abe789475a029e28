"""Row-range sharding across GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Mirrors the reference's NUMA driver (src/mat_vec.cpp:230-297): rank r plays thread/node r —
  * rows are split into `world` contiguous ranges, nrow // world each, the last takes the remainder (:233,245-246)
  * a shard keeps a rebased row_ptr and GLOBAL column indices (:250-265)
  * every rank holds a FULL replica of x (:257,266); here the replica is assembled from the ranks' own slices
    (rank r owns x[rows of r], the natural layout when y of one product feeds x of the next) by an all-gather
  * y stays sharded; concatenate_y() is the optional gather the reference only performs for DIA (:474-477)
The only exchange step is the x all-gather; there is no reduction (rows are independent).

Compute is NOT in this module: callers apply their shard with the HIP engine (capi.Context.apply).  That keeps
the collective logic testable on CPU with gloo, where the tests plug the oracle in as the per-shard product.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import capi


def shard_rows(nrow: int, world: int, rank: int) -> tuple[int, int]:
    """[begin, end) of rank's rows — the engine's own arithmetic (spmv_partition_rows)"""
    return capi.partition_rows(nrow, world, rank)


def all_bounds(nrow: int, world: int) -> list[tuple[int, int]]:
    return [shard_rows(nrow, world, r) for r in range(world)]


def allgather_x(x_full: torch.Tensor, x_own: torch.Tensor, nrow: int, group=None) -> None:
    """x_full[rows of r] <- rank r's x_own, for every r.  One collective when the slices are equal
    (all_gather_into_tensor: ring/mesh over xGMI under RCCL), one broadcast per rank otherwise."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = all_bounds(nrow, world)
    b, e = bounds[rank]
    if x_own.numel() != e - b:
        raise ValueError(f"rank {rank} owns rows [{b},{e}) but passed a slice of {x_own.numel()} entries")
    if x_full.numel() != nrow:
        raise ValueError(f"x_full has {x_full.numel()} entries, expected {nrow}")
    if x_full.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (several ranks sharing one GPU, where RCCL refuses to run): stage through the host
        host = torch.empty(nrow, dtype=x_full.dtype)
        allgather_x(host, x_own.cpu(), nrow, group)
        x_full.copy_(host)
        return
    if nrow % world == 0:
        dist.all_gather_into_tensor(x_full, x_own.contiguous(), group=group)
        return
    for r, (rb, re) in enumerate(bounds):
        part = x_full[rb:re]
        if r == rank:
            part.copy_(x_own)
        if re > rb:
            dist.broadcast(part, src=dist.get_global_rank(group, r) if group is not None else r, group=group)


def concatenate_y(y_own: torch.Tensor, nrow: int, group=None) -> torch.Tensor:
    """full y on every rank from the row slices (the reference's DIA driver copy-back, generalised)"""
    y_full = torch.empty(nrow, dtype=y_own.dtype, device=y_own.device)
    allgather_x(y_full, y_own, nrow, group)
    return y_full


def max_over_ranks(value: float, device, group=None) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
