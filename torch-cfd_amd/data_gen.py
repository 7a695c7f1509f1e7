"""Ensemble trajectory generation: the loop of the reference's data-generation drivers, device side only.

Mirrors the batch loop of fno/data_gen/data_gen_McWilliams2d.py:119-171 (IC stack -> rfft2 -> warm-up steps ->
``get_trajectory_imex`` -> irfft2 -> bilinear subsample -> dtype cast -> dict with ``random_states``) and its
final on-disk layout (``pickle_to_pt``, fno/data_gen/data_utils.py:309-328: one dict of tensors concatenated
over the batches: vorticity / stream / vort_t / residual of shape (N, T, ns, ns) + int32 ``random_states``).
The argparse / logging / dill-append plumbing of the drivers is out of scope.

Everything between the seeded CPU noise and the final ``.cpu()`` runs on the GPU: IC generation, the fused
RK4-CN steps (the warm-up is ONE multi-step library call), the record sweeps and the c2r of the records (HIP
irfft2) with the bilinear subsample inside its row pass for power-of-two factors (a torch device op otherwise).  With a process group initialised every rank generates a
contiguous slice of the samples and rank ``dst`` receives the concatenated dict (``distributed.gather_trajectory``).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .distributed import TRAJECTORY_FIELDS
from .equations import NavierStokes2DSpectral, RK4CrankNicolsonStepper, fft_plan
from .grids import Grid
from .initial_conditions import vorticity_field
from .solvers import get_trajectory_imex


DATASET_FIELDS = TRAJECTORY_FIELDS + ("random_states",)


def spectral_to_physical(value_hat: torch.Tensor, out_size: Optional[int] = None,
                         dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """(..., n, m) half spectra -> (..., ns, ns) real fields: HIP irfft2, then (if ns != n) bilinear
    interpolation exactly as ``F.interpolate(value, size=(ns, ns), mode='bilinear')`` in the drivers
    (fno/data_gen/data_gen_McWilliams2d.py:158-163) -- one fused pass when n / ns is a power of two."""
    n = value_hat.shape[-2]
    plan = fft_plan(n, torch.promote_types(value_hat.dtype, torch.complex64), value_hat.device)
    factor = plan.subsample_factor(out_size) if hasattr(plan, "subsample_factor") else 0
    if factor and plan.rdtype == dtype and os.environ.get("TCFD_FUSED_SUBSAMPLE", "1") != "0":
        # power-of-two factor: c2r + subsample as one pass (tcfd_irfft2_subsample); same result as the two steps below
        # (bit for bit at factor 2, to rounding beyond)
        return plan.irfft2_subsample(value_hat, factor)
    phys = plan.irfft2(value_hat).to(dtype)
    if out_size is not None and out_size != n:
        lead = phys.shape[:-2]
        phys = F.interpolate(phys.reshape(-1, 1, n, n), size=(out_size, out_size), mode="bilinear").reshape(
            *lead, out_size, out_size)
    return phys


def generate_mcwilliams_dataset(n: int, total_samples: int, batch_size: int, dt: float, warmup_steps: int,
                                total_steps: int, record_every_steps: int, viscosity: float = 1e-3,
                                diam: float = 2 * torch.pi, peak_wavenumber: float = 4, random_state: int = 0,
                                subsample: int = 1, dtype: torch.dtype = torch.float32,
                                cdtype: torch.dtype = torch.complex64, device="cuda", dst: int = 0,
                                path: Optional[str] = None, stats: Optional[dict] = None,
                                as_rank0_of: Optional[int] = None) -> Optional[Dict[str, torch.Tensor]]:
    """Decaying-turbulence ensemble (McWilliams IC).  Computes in the torch default dtype (the reference driver
    sets float64), stores ``dtype`` / ``cdtype``.  Returns the dataset dict on rank ``dst`` (saved with
    ``torch.save`` when ``path`` is given), ``None`` on the other ranks.

    Every record is post-processed (c2r, subsample, cast) on the device as soon as it exists and handed to the host
    of ``dst`` on side streams while the next steps run (``distributed.RecordHandover``): peers -> ``dst`` over RCCL,
    ``dst`` -> its page-locked result over PCIe.  ``stats`` (optional dict) receives the wall-clock split: ``setup_s``
    (operator, plans, page-locked result), ``stepping_s`` (until this rank's last step has finished on the device,
    hand-over of all earlier records running underneath) and ``handover_tail_s`` (what is left of the hand-over
    after that: the un-hidden part of gather + D2H).

    ``as_rank0_of=N`` (single process only) runs what rank 0 of an N-rank job runs -- the full-size page-locked result, rank
    0's batches -- with nobody to receive from: the one-GPU PROXY of the N-GPU job's wall time that ``bench.py`` reports
    (the rows of the other ranks stay uninitialised; what the proxy leaves out is the receive + D2H of their records on the
    side streams)."""
    def make_operator(grid):
        return NavierStokes2DSpectral(viscosity=viscosity, grid=grid, drag=0, smooth=True, forcing_fn=None,
                                      solver=RK4CrankNicolsonStepper())

    def initial_vorticity(grid, start, count, device):
        return vorticity_field(grid, peak_wavenumber, batch_seeds=[random_state + start + k for k in range(count)], device=device)

    return _generate_dataset(n, total_samples, batch_size, dt, warmup_steps, total_steps, record_every_steps, make_operator,
                             initial_vorticity, diam, random_state, subsample, dtype, cdtype, device, dst, path, stats, as_rank0_of)


def generate_kolmogorov_dataset(n: int, total_samples: int, batch_size: int, dt: float, warmup_steps: int,
                                total_steps: int, record_every_steps: int, viscosity: float = 1e-3,
                                diam: float = 2 * torch.pi, peak_wavenumber: float = 4, max_velocity: float = 5,
                                scale: float = 1.0, drag: float = 0.1, random_state: int = 0, subsample: int = 1,
                                dtype: torch.dtype = torch.float32, cdtype: torch.dtype = torch.complex64, device="cuda",
                                dst: int = 0, path: Optional[str] = None, stats: Optional[dict] = None,
                                as_rank0_of: Optional[int] = None) -> Optional[Dict[str, torch.Tensor]]:
    """Forced-turbulence ensemble: the loop of fno/data_gen/data_gen_Kolmogorov2d.py:119-192 -- Kolmogorov forcing
    ``scale * sin(peak_wavenumber * y)`` (``:121-126``), drag 0.1, RK4-CN (``:128-135``), initial vorticity
    ``curl_2d(filtered_velocity_field(grid, max_velocity, peak_wavenumber, random_state=...))`` per sample (``:144-155``),
    ``warmup_steps`` unrecorded steps (``:158-168``), then ``get_trajectory_imex`` with a record every
    ``record_every_steps`` (``:170-177``), irfft2 -> bilinear subsample -> cast (``:179-189``), ``random_states`` (``:191-193``).
    Same dataset dict, hand-over and multi-rank split as ``generate_mcwilliams_dataset``.

    Seeds as the reference draws them: sample k of batch i is generated from ``random_state + i + k`` -- the batch INDEX, not
    the sample offset ``idx`` (``:151``) -- while ``random_states`` records ``random_state + idx + k`` (``:192``); with more
    than one sample per batch consecutive batches therefore share initial conditions.  Restated, not corrected; the batch
    index of a sample is that of the serial loop (global index // batch_size) however the samples are cut across ranks."""
    from .forcings import KolmogorovForcing
    from .initial_conditions import curl_2d, filtered_velocity_field

    def make_operator(grid):
        forcing = KolmogorovForcing(grid=grid, scale=scale, wave_number=peak_wavenumber, swap_xy=False)
        return NavierStokes2DSpectral(viscosity=viscosity, grid=grid, drag=drag, smooth=True, forcing_fn=forcing,
                                      solver=RK4CrankNicolsonStepper())

    def initial_vorticity(grid, start, count, device):
        seeds = [random_state + g // batch_size + g % batch_size for g in range(start, start + count)]
        return curl_2d(filtered_velocity_field(grid, max_velocity, peak_wavenumber, batch_seeds=seeds, device=device), grid)

    return _generate_dataset(n, total_samples, batch_size, dt, warmup_steps, total_steps, record_every_steps, make_operator,
                             initial_vorticity, diam, random_state, subsample, dtype, cdtype, device, dst, path, stats, as_rank0_of)


def _generate_dataset(n: int, total_samples: int, batch_size: int, dt: float, warmup_steps: int, total_steps: int,
                      record_every_steps: int, make_operator, initial_vorticity, diam: float, random_state: int,
                      subsample: int, dtype: torch.dtype, cdtype: torch.dtype, device, dst: int, path: Optional[str],
                      stats: Optional[dict], as_rank0_of: Optional[int]) -> Optional[Dict[str, torch.Tensor]]:
    """The batch loop both drivers share (fno/data_gen/data_gen_McWilliams2d.py:119-171, data_gen_Kolmogorov2d.py:134-192):
    ``make_operator(grid)`` builds the equation, ``initial_vorticity(grid, start, count, device)`` the (count, n, n) physical
    initial vorticity of the samples with global indices start .. start + count - 1."""
    import time

    import torch.distributed as dist

    from .distributed import RecordHandover, batch_layout

    t_begin = time.perf_counter()
    device = torch.device(device)
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if distributed else 1
    rank = dist.get_rank() if distributed else 0
    layout = batch_layout(total_samples, world, batch_size)
    if as_rank0_of is not None:
        if distributed and world > 1:
            raise ValueError("as_rank0_of is the single-process proxy of a multi-rank job")
        layout = [batch_layout(total_samples, as_rank0_of, batch_size)[0]]
    grid = Grid(shape=(n, n), domain=((0, diam), (0, diam)), device=device)
    op = make_operator(grid).to(device)
    real = torch.get_default_dtype()
    plan = fft_plan(n, torch.complex128 if real == torch.float64 else torch.complex64, device, diam)
    ns = n // subsample
    n_rec = len(range(0, total_steps, record_every_steps))
    handover = RecordHandover(TRAJECTORY_FIELDS, total_samples, n_rec, (ns, ns), dtype, layout, device, dst=dst)
    on_gpu = device.type == "cuda"
    if on_gpu and layout[rank]:
        # every plan the loop will ask for, NOW: plan creation is host work on fresh memory (tables in float64 on the CPU,
        # their upload), and it must not run beside the page-locking of the result -- that holds the process's memory-map
        # lock and makes every page fault of this thread wait (measured: the first record of the C4 job 0.14 s late)
        cplx = torch.complex128 if real == torch.float64 else torch.complex64
        op._plan(torch.empty(0, n, n // 2 + 1, dtype=cplx, device=device))
        fft_plan(n, torch.promote_types(cdtype, torch.complex64), device)
    if on_gpu:
        torch.cuda.synchronize(device)
    t_setup = time.perf_counter()
    try:
        for start, count in layout[rank]:
            w = plan.rfft2(initial_vorticity(grid, start, count, device))
            handover.start_allocation()     # page-lock the result now: under the warm-up steps, not under the CPU noise above
            if warmup_steps > 0:
                w, _ = op._fused_steps(w, dt, warmup_steps, want_dwdt=False)

            def sink(rec, fields, start=start, count=count):
                packed = torch.empty((count, len(TRAJECTORY_FIELDS), ns, ns), dtype=dtype, device=device)
                for f, name in enumerate(TRAJECTORY_FIELDS):
                    packed[:, f] = spectral_to_physical(fields[name], ns, dtype)
                handover.push(start, rec, packed)

            get_trajectory_imex(op, w, dt, num_steps=total_steps, record_every_steps=record_every_steps, dtype=cdtype,
                                to_cpu=False, record_sink=sink)
        if on_gpu:
            torch.cuda.current_stream(device).synchronize()
        t_stepped = time.perf_counter()
        full = handover.finish()
    except BaseException:
        handover.close()      # stop the page-locking helper, wait for copies in flight, unlock: the result is being dropped
        raise
    t_end = time.perf_counter()
    if stats is not None and handover.trace is not None:
        stats["trace"] = list(handover.trace)
    if stats is not None:
        stats.update(setup_s=t_setup - t_begin, stepping_s=t_stepped - t_setup, handover_tail_s=t_end - t_stepped,
                     batches=len(layout[rank]), samples=sum(c for _, c in layout[rank]), handover_mode=handover.mode)
    if full is None:
        return None
    full = dict(full)
    full["random_states"] = torch.arange(random_state, random_state + total_samples, dtype=torch.int32)
    if path is not None:
        torch.save(full, path)
    return full
