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
from typing import Callable, Dict, Optional

import torch

from .equations import ImplicitExplicitODE, _is_module_stepper

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
    record_sink: Optional[Callable[[int, Dict[str, torch.Tensor]], None]] = None,
) -> Dict[str, torch.Tensor]:
    """w0 (*, n, m) half spectrum -> dict(vorticity, stream, vort_t, residual),
    each (*, n_t, n, m) complex ``dtype`` with one snapshot after every step whose
    0-based index is a multiple of ``record_every_steps``.  ``to_cpu=False`` keeps
    the stacks on the device (used by the multi-GPU gather).  With a ``record_sink`` nothing is stacked: every
    record goes to ``record_sink(index, {name: (*, n, m) complex dtype})`` as soon as it exists (device tensors on the
    current stream) and the function returns ``{}`` -- the ensemble driver post-processes and hands records over while
    the following steps run (``data_gen.py``, ``distributed.RecordHandover``)."""
    # require_grad: the reference marks w / dw/dt of every step as requiring grad and records DETACHED copies
    # (solvers.py:224-250); here the steps then run through the differentiable path (torch-cfd_amd/autograd.py)
    if require_grad:
        w0 = w0.detach().requires_grad_(True)
    n_rec = len(range(0, num_steps, record_every_steps))
    lead, (n, m) = tuple(w0.shape[:-2]), w0.shape[-2:]
    names = ("vorticity", "stream", "vort_t", "residual")
    out = {} if record_sink is not None else {k: torch.empty(*lead, n_rec, n, m, dtype=dtype, device=w0.device)
                                              for k in names}
    tqdm_iters = num_steps if TQDM_ITERS > num_steps else TQDM_ITERS
    update_every = max(num_steps // tqdm_iters, 1)
    bar = None
    if pbar:
        from tqdm import tqdm

        bar = tqdm(total=num_steps)
    w = w0
    rec = 0
    fused = (bar is None and not require_grad and hasattr(equation, "_fused_steps")
             and _is_module_stepper(getattr(equation, "solver", None)))
    t_step = 0
    while t_step < num_steps:
        if fused and t_step % record_every_steps != 0:
            # the steps strictly between two records need no dw/dt and no host round trip: one library call
            nxt = min((t_step // record_every_steps + 1) * record_every_steps, num_steps)
            w, _ = equation._fused_steps(w, dt, nxt - t_step, want_dwdt=False)
            t_step = nxt
            continue
        with torch.set_grad_enabled(bool(require_grad)):
            w, dwdt = equation.forward(w, dt=dt)
        if bar is not None and t_step % update_every == 0:
            res = equation.residual(w, dwdt)
            res_norm = torch.linalg.norm(res.reshape(-1, n * m), dim=-1).mean().item() / n
            bar.set_description(f"{datetime.now():%d-%b-%Y %H:%M:%S} - {pbar_desc} - ||L(w) - f||: {res_norm:.4e}")
            bar.update(update_every)
        if t_step % record_every_steps == 0:
            psi, res = equation.stream_and_residual(w.detach(), dwdt.detach())
            if record_sink is not None:
                record_sink(rec, {key: val.to(dtype) for key, val in zip(names, (w.detach(), psi, dwdt.detach(), res))})
            else:
                for key, val in zip(names, (w.detach(), psi, dwdt.detach(), res)):
                    out[key][..., rec, :, :].copy_(val)  # casts to `dtype` on the device
            rec += 1
        t_step += 1
    if bar is not None:
        bar.close()
    if to_cpu:
        out = {k: v.cpu() for k, v in out.items()}
    return out


# ----------------------------------------------------------------------------- legacy IMEX Crank-Nicolson step
_CN_PLANS: Dict[tuple, object] = {}


def _convection_hat(w: torch.Tensor, kx: torch.Tensor, ky: torch.Tensor, mask: torch.Tensor):
    """rfft2(u w_x + v w_y) (times ``mask``) = minus the de-aliased advection term of the RK operator: the same
    three HIP kernels (column pass, fused row pass, column pass) with a plan that has no linear term / forcing."""
    from .equations import _COMPLEX_OF, _HipPlan

    n, m = w.shape[-2:]
    cdtype = torch.promote_types(w.dtype, _COMPLEX_OF.get(kx.dtype, torch.complex64))
    kx1 = kx.reshape(-1, n, m)[0, :, 0].detach().cpu().double()
    ky1 = ky.reshape(-1, n, m)[0, 0, :].detach().cpu().double()
    mk = mask.reshape(-1, n, m)[0].detach().cpu().double()
    key = (n, cdtype, w.device, float(kx1[1]), float(ky1[1]), float(mk.sum()), int(mk[:, 0].sum()), int(mk[0, :].sum()))
    plan = _CN_PLANS.get(key)
    if plan is None:
        plan = _HipPlan(n, cdtype, w.device, kx1, ky1, torch.zeros(n, m), mk)
        _CN_PLANS[key] = plan
    return -plan.explicit_terms(w).reshape(w.shape)


def _cn_tables(w, diam, rfftmesh, laplacian, dealias_filter):
    n = w.shape[-2]
    k_max = math.floor(n / 2.0)
    real = torch.float64 if w.dtype == torch.complex128 else torch.float32
    if rfftmesh is None:
        k = torch.fft.fftfreq(n, d=diam / n)
        kx, ky = torch.meshgrid([k, k], indexing="ij")
        kx, ky = kx[..., : k_max + 1], ky[..., : k_max + 1]
    else:
        kx, ky = rfftmesh
    kx, ky = [z.to(real).to(w.device) for z in (kx, ky)]
    if laplacian is None:
        laplacian = -4 * (math.pi**2) * (kx**2 + ky**2)
        laplacian[..., 0, 0] = 1.0
    if dealias_filter is None:
        dealias_filter = torch.logical_and(torch.abs(ky) <= (2.0 / 3.0) * k_max, torch.abs(kx) <= (2.0 / 3.0) * k_max)
    return kx, ky, laplacian.to(w.device), dealias_filter.to(w.device)


def update_residual(w_h, w_h_t, f_h, visc, rfftmesh, laplacian, dealias_filter=None, dealias=True, **kwargs):
    """Residual  w_t + u.grad(w) - nu lap w - f  in Fourier space (fno/data_gen/solvers.py:49-88)."""
    kx, ky = rfftmesh
    use = dealias and dealias_filter is not None
    mask = dealias_filter.to(kx.dtype) if use else torch.ones_like(kx)
    conv = _convection_hat(w_h, kx, ky, mask)
    return w_h_t + conv - visc * laplacian * w_h - f_h


def imex_crank_nicolson_step(w, f, visc, delta_t, diam: float = 1, rfftmesh=None, laplacian=None,
                             dealias_filter=None, dealias: bool = False, output_rfft: bool = False, debug=False,
                             **kwargs):
    """One first-order IMEX step with Crank-Nicolson diffusion (fno/data_gen/solvers.py:91-188, the stepper
    behind the fine-tuning head).  Returns (w_next, dw/dt, w, psi_h, residual[, mesh, laplacian, filter])."""
    size = w.shape[1:]
    assert (size[-1] - 1) * 2 == size[-2]  # an rfft2 tensor
    kx, ky, laplacian, dealias_filter = _cn_tables(w, diam, rfftmesh, laplacian, dealias_filter)
    if f.ndim < w.ndim:
        f = f.unsqueeze(0)
    f = f.to(w.device)
    psi_h = -w / laplacian
    mask = dealias_filter.to(kx.dtype) if dealias else torch.ones_like(kx)
    convection_h = _convection_hat(w, kx, ky, mask)
    half = 0.5 * delta_t * visc * laplacian
    w_next = (-delta_t * convection_h + delta_t * f + (1.0 + half) * w) / (1.0 - half)
    dwdt = (w_next - w) / delta_t
    res_h = dwdt + convection_h - visc * laplacian * w - f
    if output_rfft:
        return w_next, dwdt, w, psi_h, res_h, (kx, ky), laplacian, dealias_filter
    return w_next, dwdt, w, psi_h, res_h


# ----------------------------------------------------------------------------- legacy driver around that step
_BDF_WEIGHTS = {
    1: [1, -1],
    2: [3 / 2, -2, 0.5],
    3: [11 / 6, -3, 3 / 2, -1 / 3],
    4: [25 / 12, -4, 3, -4 / 3, 1 / 4],
    5: [137 / 60, -5, 5, -10 / 3, 5 / 4, -1 / 5],
}


def backdiff(x: torch.Tensor, order: int = 3) -> torch.Tensor:
    """Backward-difference (BDF) combination of the last ``order + 1`` time samples of x (b, *, x, y, t): the unscaled
    time derivative at the last sample (fno/data_gen/solvers.py:19-35; weights in the default dtype, newest sample first)."""
    if order > 5:
        raise NotImplementedError("only bdf order <= 5 is implemented")
    weights = torch.as_tensor(_BDF_WEIGHTS[order]).to(x.device)
    return (x[..., -(order + 1):].flip(-1) * weights).sum(-1)


def get_trajectory_imex_crank_nicolson(w0: torch.Tensor, f: torch.Tensor, visc: float = 1e-3, T: float = 1, delta_t: float = 1e-3,
                                       record_steps: int = 1, diam: float = 1, dealias: bool = True, subsample: int = 1,
                                       dtype: Optional[torch.dtype] = None, pbar: bool = True, **kwargs) -> Dict[str, torch.Tensor]:
    """The legacy data-generation loop (fno/data_gen/solvers.py:268-448; caller fno/data_gen/data_gen_fno_legacy.py:181-207):
    ``ceil(T / delta_t)`` first-order IMEX Crank-Nicolson steps from the PHYSICAL initial vorticity w0 (B, n, n) under the
    fixed physical forcing f ((n, n) or (B, n, n)), a record every ``floor(total_steps / record_steps)`` steps of
    {vorticity, its time derivative, stream function, PDE residual (``update_residual`` of the NEW state with the step's
    dw/dt)} in physical space, bilinearly subsampled to n // subsample (``F.interpolate(size=, mode="bilinear")``, :37-46).
    Returns ``dict(vorticity, vorticity_t, stream, residual: (B, record_steps, ns, ns); t_steps: (record_steps,))`` on the
    CPU in the default dtype, as the reference allocates them.

    Device side: r2c / c2r are the HIP transforms, the convection term of every step and of every record's residual the
    fused column / row / column kernels (``_convection_hat``); records are subsampled inside the c2r row pass for
    power-of-two factors and copied to the host once at the end.  The reference tests the state for NaNs after EVERY step
    (one host round trip each, :386-388); here the test runs when a record is taken and after the last step, with the same
    ``ValueError``.  A schedule that would produce more records than ``record_steps`` fails with the reference's
    ``IndexError`` (its ``vort[:, c] = ...`` past the end), raised before any step is taken."""
    from .data_gen import spectral_to_physical
    from .equations import _COMPLEX_OF, fft_plan

    real = w0.dtype if dtype is None else dtype
    device = w0.device
    bsz, n = w0.size(0), w0.size(-1)
    ns = n // subsample
    k_max = math.floor(n / 2.0)
    total_steps = math.ceil(T / delta_t)
    record_every_n_steps = math.floor(total_steps / record_steps)
    if record_every_n_steps < 1 or total_steps // record_every_n_steps > record_steps:
        raise IndexError(f"{total_steps} steps with a record every {record_every_n_steps} give "
                         f"{total_steps // max(record_every_n_steps, 1)} records for {record_steps} slots")
    plan = fft_plan(n, _COMPLEX_OF[w0.dtype], device, diam)
    w_h = plan.rfft2(w0.contiguous())
    f_h = plan.rfft2(f.to(device=device, dtype=w0.dtype).contiguous().reshape(-1, n, n)).reshape(*f.shape[:-2], n, n // 2 + 1)
    if f_h.ndim < w_h.ndim:
        f_h = f_h.unsqueeze(0)
    k = torch.fft.fftfreq(n, d=diam / n, dtype=real, device=device)
    kx, ky = torch.meshgrid([k, k], indexing="ij")
    kx, ky = kx[..., : k_max + 1], ky[..., : k_max + 1]
    k_cut = (1 / diam) * k_max
    lap = -4 * (math.pi**2) * (kx**2 + ky**2)
    lap[0, 0] = 1.0
    kx, ky, lap = kx[None, ...], ky[None, ...], lap[None, ...]
    dealias_filter = (torch.logical_and(torch.abs(kx) <= (2.0 / 3.0) * k_cut, torch.abs(ky) <= (2.0 / 3.0) * k_cut).to(real)
                      if dealias else None)
    out_real = torch.get_default_dtype()
    names = ("vorticity", "vorticity_t", "stream", "residual")
    dev_out = {key: torch.empty(bsz, record_steps, ns, ns, dtype=out_real, device=device) for key in names}
    t_steps = torch.empty(record_steps, device="cpu")
    bar = None
    if pbar:
        from tqdm import tqdm

        bar = tqdm(total=total_steps)

    def diverged(w):
        if torch.isnan(torch.view_as_real(w)).any():
            raise ValueError(f"Solution diverged with norm {torch.linalg.norm(w[~torch.isnan(w)])}")

    c, t = 0, 0.0
    for j in range(total_steps):
        w_h, w_h_t, _, psi_h, _ = imex_crank_nicolson_step(w_h, f_h, visc, delta_t, diam=diam, rfftmesh=(kx, ky), laplacian=lap,
                                                           dealias_filter=dealias_filter, dealias=dealias, **kwargs)
        t += delta_t
        if (j + 1) % record_every_n_steps == 0:
            diverged(w_h)
            res_h = update_residual(w_h, w_h_t, f_h, visc, (kx, ky), lap, dealias_filter=dealias_filter, dealias=dealias)
            for key, val in zip(names, (w_h, w_h_t, psi_h, res_h)):
                dev_out[key][:, c] = spectral_to_physical(val.contiguous(), ns, val.real.dtype)
            t_steps[c] = t
            c += 1
            if bar is not None:
                enstrophy = torch.linalg.norm(dev_out["vorticity"][:, c - 1], dim=(-1, -2)).mean().item() / n
                res_l2 = torch.linalg.norm(dev_out["residual"][:, c - 1], dim=(-1, -2)).mean().item() / n
                bar.set_description(f"{datetime.now():%d-%b-%Y %H:%M:%S} - enstrophy w: {enstrophy:.4f}  ||L(w, psi) - f||_2: {res_l2:.4e}")
        if bar is not None:
            bar.update()
    diverged(w_h)
    if bar is not None:
        bar.close()
    out = {key: val.cpu() for key, val in dev_out.items()}
    out["t_steps"] = t_steps
    return out
