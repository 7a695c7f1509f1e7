"""Batch-sharded ensembles over the GPUs of one node.

Every batch element of the spectral solver is an independent trajectory (all operators act on the last two
dims), so the path shards with NO collective inside a step: each rank owns a contiguous slice of the batch, its
own plan and workspace.  The only data-path communication is the hand-over of recorded snapshots to one rank at
the end of a trajectory: point-to-point sends straight into the slices of ONE pre-allocated result on ``dst``
(RCCL over xGMI for HIP tensors -- seven peers send over seven distinct links in parallel; gloo on CPU in the
tests).  Nobody but ``dst`` allocates anything, shards may be ragged or empty.  The reference has no distributed
code; this is what its data-generation drivers need
(fno/data_gen/data_gen_McWilliams2d.py:126-152 loops over batches serially on one device).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

TRAJECTORY_FIELDS = ("vorticity", "stream", "vort_t", "residual")


def shard_batch(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[start, stop) of ``rank``'s contiguous slice of ``total`` batch elements (sizes differ by at most 1)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(total, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _describe(local: Dict[str, torch.Tensor]) -> List[tuple]:
    return [(k, tuple(v.shape[1:]), str(v.dtype)) for k, v in sorted(local.items())]


def gather_trajectory(local: Dict[str, torch.Tensor], total_batch: int, dst: int = 0,
                      group: Optional[dist.ProcessGroup] = None,
                      keys: Optional[Sequence[str]] = None) -> Optional[Dict[str, torch.Tensor]]:
    """Concatenate per-rank dicts ``{key: (B_rank, ...)}`` along the batch axis on rank ``dst``.

    Every rank takes part with the SAME key list: ``keys`` if given, else the description (key, trailing shape,
    dtype) of the first non-empty shard, agreed on with one small object collective -- a rank whose shard is empty
    (``total_batch < world_size``) holds no tensors and still has to know what the others send.  Returns the
    assembled dict on ``dst`` and ``None`` elsewhere; without a process group the input is returned unchanged."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    spans = [shard_batch(total_batch, r, world) for r in range(world)]
    lo, hi = spans[rank]
    mine = _describe(local) if keys is None else _describe({k: local[k] for k in keys if k in local})
    every: List[Optional[list]] = [None] * world
    dist.all_gather_object(every, mine, group=group)
    layout = next((d for d in every if d), None)
    if layout is None:
        return {} if rank == dst else None
    for r, d in enumerate(every):
        if d and d != layout:
            raise ValueError(f"rank {r} holds {d}, rank layout agreed on is {layout}")
    some = next(iter(local.values())) if local else None
    device = some.device if some is not None else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    out: Dict[str, torch.Tensor] = {}
    ops, keep = [], []
    for key, trailing, dtype_name in layout:
        dtype = getattr(torch, dtype_name.replace("torch.", ""))
        x = local.get(key)
        if x is not None and x.shape[0] != hi - lo:
            raise ValueError(f"{key}: local batch {x.shape[0]} != shard size {hi - lo}")
        if rank == dst:
            full = torch.empty((total_batch,) + trailing, dtype=dtype, device=device)
            out[key] = full
            if hi > lo:
                full[lo:hi].copy_(x)
            for r, (a, b) in enumerate(spans):
                if r != dst and b > a:
                    view = full[a:b]   # contiguous: the slice is along the leading axis
                    ops.append(dist.P2POp(dist.irecv, torch.view_as_real(view) if view.is_complex() else view, r,
                                          group=group))
        elif hi > lo:
            xs = x.contiguous()
            keep.append(xs)
            ops.append(dist.P2POp(dist.isend, torch.view_as_real(xs) if xs.is_complex() else xs, dst, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out if rank == dst else None


def sharded_trajectory(equation, w0_local: torch.Tensor, total_batch: int, dt: float, num_steps: int,
                       record_every_steps: int = 1, dtype: torch.dtype = torch.complex64, dst: int = 0,
                       group: Optional[dist.ProcessGroup] = None):
    """Run ``get_trajectory_imex`` on this rank's shard (records stay on the device) and hand them to ``dst``."""
    from .solvers import get_trajectory_imex

    local = get_trajectory_imex(equation, w0_local, dt, num_steps=num_steps, record_every_steps=record_every_steps,
                                dtype=dtype, to_cpu=False) if w0_local.shape[0] > 0 else {}
    full = gather_trajectory(local, total_batch, dst=dst, group=group, keys=TRAJECTORY_FIELDS)
    if full is not None and dist.is_initialized() and dist.get_backend(group) != "gloo":
        full = {k: v.cpu() for k, v in full.items()}
    return full
