"""Trajectory recorder of the data-generation drivers (fno/data_gen/solvers.py:191-265).

Same signature and return value as ``get_trajectory_imex``; differences are all
on the device side:
  * the step is the fused HIP RK4-CN step,
  * psi and the PDE residual of a record come from ONE fused sweep
    (``tcfd_ns2d_stream_residual``) instead of vorticity_to_velocity + residual,
  * records are written into pre-allocated device tensors and copied to the host
    once at the end instead of 4 blocking ``.cpu()`` calls per record,
  * the progress-bar residual (one extra F evaluation + host sync per tick in the
    reference, solvers.py:228-243) is only computed when ``pbar=True``.
"""
from __future__ import annotations

import math
from datetime import datetime
from typing import Dict

import torch

from .equations import ImplicitExplicitODE, RK4CrankNicolsonStepper

TQDM_ITERS = 200


def get_trajectory_imex(
    equation: ImplicitExplicitODE,
    w0: torch.Tensor,
    dt: float,
    num_steps: int = 1,
    record_every_steps: int = 1,
    pbar: bool = False,
    pbar_desc: str = "generating trajectories using RK4",
    require_grad: bool = False,
    dtype: torch.dtype = torch.complex64,
    to_cpu: bool = True,
) -> Dict[str, torch.Tensor]:
    """w0 (*, n, m) half spectrum -> dict(vorticity, stream, vort_t, residual),
    each (*, n_t, n, m) complex ``dtype`` with one snapshot after every step whose
    0-based index is a multiple of ``record_every_steps``.  ``to_cpu=False`` keeps
    the stacks on the device (used by the multi-GPU gather)."""
    if require_grad:
        raise NotImplementedError("the HIP spectral path is forward-only (require_grad=True unsupported)")
    n_rec = len(range(0, num_steps, record_every_steps))
    lead, (n, m) = tuple(w0.shape[:-2]), w0.shape[-2:]
    names = ("vorticity", "stream", "vort_t", "residual")
    out = {k: torch.empty(*lead, n_rec, n, m, dtype=dtype, device=w0.device) for k in names}
    tqdm_iters = num_steps if TQDM_ITERS > num_steps else TQDM_ITERS
    update_every = max(num_steps // tqdm_iters, 1)
    bar = None
    if pbar:
        from tqdm import tqdm

        bar = tqdm(total=num_steps)
    w = w0
    rec = 0
    fused = bar is None and isinstance(getattr(equation, "solver", None), RK4CrankNicolsonStepper)
    t_step = 0
    while t_step < num_steps:
        if fused and t_step % record_every_steps != 0:
            # the steps strictly between two records need no dw/dt and no host round trip: one library call
            nxt = min((t_step // record_every_steps + 1) * record_every_steps, num_steps)
            w, _ = equation._fused_steps(w, dt, nxt - t_step, want_dwdt=False)
            t_step = nxt
            continue
        w, dwdt = equation.forward(w, dt=dt)
        if bar is not None and t_step % update_every == 0:
            res = equation.residual(w, dwdt)
            res_norm = torch.linalg.norm(res.reshape(-1, n * m), dim=-1).mean().item() / n
            bar.set_description(f"{datetime.now():%d-%b-%Y %H:%M:%S} - {pbar_desc} - ||L(w) - f||: {res_norm:.4e}")
            bar.update(update_every)
        if t_step % record_every_steps == 0:
            psi, res = equation.stream_and_residual(w, dwdt)
            for key, val in zip(names, (w, psi, dwdt, res)):
                out[key][..., rec, :, :].copy_(val)  # casts to `dtype` on the device
            rec += 1
        t_step += 1
    if bar is not None:
        bar.close()
    if to_cpu:
        out = {k: v.cpu() for k, v in out.items()}
    return out
