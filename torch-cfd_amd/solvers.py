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
# backward-difference weights of order 1 ... 5, newest sample first (fno/data_gen/solvers.py:26-32)
_BDF_WEIGHTS = {1: (1.0, -1.0), 2: (1.5, -2.0, 0.5), 3: (11 / 6, -3.0, 1.5, -1 / 3), 4: (25 / 12, -4.0, 3.0, -4 / 3, 0.25),
                5: (137 / 60, -5.0, 5.0, -10 / 3, 1.25, -0.2)}


def backdiff(x: torch.Tensor, order: int = 3) -> torch.Tensor:
    """Backward-difference (BDF) combination of the last ``order + 1`` time samples of x (b, *, x, y, t): the unscaled
    time derivative at the last sample (fno/data_gen/solvers.py:19-35; weights in the default dtype, newest sample first)."""
    if order not in _BDF_WEIGHTS:
        if order > 5:
            raise NotImplementedError("only bdf order <= 5 is implemented")
        raise KeyError(order)
    newest_first = x[..., -(order + 1):].flip(-1)
    return (newest_first * torch.as_tensor(_BDF_WEIGHTS[order]).to(x.device)).sum(-1)


def _legacy_tables(n: int, diam: float, real: torch.dtype, device, dealias: bool):
    """Wavenumber mesh, patched Laplacian and 2/3 mask of the legacy driver, each with a leading broadcast axis
    (fno/data_gen/solvers.py:316-345: the mask compares |k| with 2/3 of the LARGEST wavenumber floor(n / 2) / diam)."""
    half = n // 2
    freq = torch.fft.fftfreq(n, d=diam / n, dtype=real, device=device)
    kx = freq[:, None].expand(n, half + 1)[None]
    ky = freq[None, : half + 1].expand(n, half + 1)[None]      # (slot n/2 holds -n/2 / diam, as the reference's sliced meshgrid does)
    lap = (-4 * math.pi**2) * (kx * kx + ky * ky)
    lap = lap.clone()
    lap[0, 0, 0] = 1.0
    mask = None
    if dealias:
        cut = (2.0 / 3.0) * (half / diam)
        mask = ((kx.abs() <= cut) & (ky.abs() <= cut)).to(real)
    return kx.contiguous(), ky.contiguous(), lap, mask


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
    device, batch, n = w0.device, w0.shape[0], w0.shape[-1]
    ns = n // subsample
    n_steps = math.ceil(T / delta_t)
    stride = n_steps // record_steps if record_steps > 0 else 0       # steps between two records
    n_records = n_steps // stride if stride > 0 else 0
    if stride < 1 or n_records > record_steps:
        raise IndexError(f"{n_steps} steps with a record every {stride} give {n_records} records for {record_steps} slots")
    plan = fft_plan(n, _COMPLEX_OF[w0.dtype], device, diam)
    w_h = plan.rfft2(w0.contiguous())
    f_h = plan.rfft2(f.to(device=device, dtype=w0.dtype).contiguous().reshape(-1, n, n)).reshape(*f.shape[:-2], n, n // 2 + 1)
    if f_h.dim() == w_h.dim() - 1:
        f_h = f_h[None]
    kx, ky, lap, mask = _legacy_tables(n, diam, real, device, dealias)
    names = ("vorticity", "vorticity_t", "stream", "residual")
    records = {key: torch.empty(batch, record_steps, ns, ns, dtype=torch.get_default_dtype(), device=device) for key in names}
    times = torch.empty(record_steps, device="cpu")
    bar = None
    if pbar:
        from tqdm import tqdm

        bar = tqdm(total=n_steps)

    def check_finite(w):
        bad = torch.isnan(torch.view_as_real(w)).any()
        if bad:
            raise ValueError(f"Solution diverged with norm {torch.linalg.norm(w[~torch.isnan(w)])}")

    clock = [0.0]                            # physical time, accumulated step by step like the reference's `t += delta_t`

    def advance(w, count):
        out = None
        for _ in range(count):
            out = imex_crank_nicolson_step(w, f_h, visc, delta_t, diam=diam, rfftmesh=(kx, ky), laplacian=lap,
                                           dealias_filter=mask, dealias=dealias, **kwargs)
            w = out[0]
            clock[0] += delta_t
            if bar is not None:
                bar.update()
        return w, out

    for slot in range(n_records):
        w_h, last = advance(w_h, stride)
        check_finite(w_h)
        dw_h, psi_h = last[1], last[3]
        res_h = update_residual(w_h, dw_h, f_h, visc, (kx, ky), lap, dealias_filter=mask, dealias=dealias)
        for key, field in zip(names, (w_h, dw_h, psi_h, res_h)):
            records[key][:, slot] = spectral_to_physical(field.contiguous(), ns, field.real.dtype)
        times[slot] = clock[0]
        if bar is not None:
            enstrophy = torch.linalg.norm(records["vorticity"][:, slot], dim=(-1, -2)).mean().item() / n
            res_l2 = torch.linalg.norm(records["residual"][:, slot], dim=(-1, -2)).mean().item() / n
            bar.set_description(f"{datetime.now():%d-%b-%Y %H:%M:%S} - enstrophy w: {enstrophy:.4f}  ||L(w, psi) - f||_2: {res_l2:.4e}")
    if n_steps - n_records * stride > 0:
        w_h, _ = advance(w_h, n_steps - n_records * stride)        # the steps after the last record (their state is not returned)
    check_finite(w_h)
    if bar is not None:
        bar.close()
    result = {key: val.cpu() for key, val in records.items()}
    result["t_steps"] = times
    return result
