"""Batch-sharded ensembles over the GPUs of one node.

Every batch element of the spectral solver is an independent trajectory (all operators act on the last two
dims), so the path shards with NO collective inside a step: each rank owns a contiguous slice of the batch, its
own plan and workspace.  The only data-path collective is the gather of recorded snapshots at the end of a
trajectory (RCCL over xGMI when the tensors live on HIP devices, gloo on CPU in the tests).  The reference has
no distributed code at all; this mirrors what its data-generation drivers would need
(fno/data_gen/data_gen_McWilliams2d.py:126-152 loops over batches serially on one device).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_batch(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[start, stop) of ``rank``'s contiguous slice of ``total`` batch elements (sizes differ by at most 1)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(total, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_trajectory(local: Dict[str, torch.Tensor], total_batch: int, dst: int = 0,
                      group: Optional[dist.ProcessGroup] = None) -> Optional[Dict[str, torch.Tensor]]:
    """Concatenate per-rank trajectory dicts {(B_rank, T, n, m)} along the batch axis on rank ``dst``.

    One ``all_gather`` per field on a shard padded to the largest shard size (RCCL has no ragged gather);
    returns the assembled dict on ``dst`` and ``None`` elsewhere.  Without an initialised process group the
    input is returned unchanged (single-GPU run)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_batch(total_batch, r, world) for r in range(world)]
    bmax = max(b - a for a, b in sizes)
    out: Dict[str, torch.Tensor] = {}
    for key in sorted(local):
        x = local[key].contiguous()
        if x.shape[0] != sizes[rank][1] - sizes[rank][0]:
            raise ValueError(f"{key}: local batch {x.shape[0]} != shard size {sizes[rank][1] - sizes[rank][0]}")
        if x.shape[0] < bmax:
            pad = torch.zeros((bmax - x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            x = torch.cat([x, pad])
        # complex tensors travel as their real views (backends differ in complex support)
        xr = torch.view_as_real(x) if x.is_complex() else x
        parts = [torch.empty_like(xr) for _ in range(world)]
        dist.all_gather(parts, xr, group=group)
        if rank == dst:
            parts = [torch.view_as_complex(p) if x.is_complex() else p for p in parts]
            out[key] = torch.cat([p[: b - a] for p, (a, b) in zip(parts, sizes)])
    return out if rank == dst else None


def sharded_trajectory(equation, w0_local: torch.Tensor, total_batch: int, dt: float, num_steps: int,
                       record_every_steps: int = 1, dtype: torch.dtype = torch.complex64, dst: int = 0,
                       group: Optional[dist.ProcessGroup] = None):
    """Run ``get_trajectory_imex`` on this rank's shard (records stay on the device) and gather on ``dst``."""
    from .solvers import get_trajectory_imex

    local = get_trajectory_imex(equation, w0_local, dt, num_steps=num_steps, record_every_steps=record_every_steps,
                                dtype=dtype, to_cpu=False)
    full = gather_trajectory(local, total_batch, dst=dst, group=group)
    if full is not None and dist.is_initialized() and dist.get_backend(group) != "gloo":
        full = {k: v.cpu() for k, v in full.items()}
    return full
