"""Shows that oracle/ns2d.py is a valid wall-time proxy for the reference's CPU path ("the builder must
first show, in this container, that its wall-time matches the imported reference's within noise").

Runs in the BUILD container only (it imports /root/reference); nothing on the GPU box uses it.  Times
NavierStokes2DSpectral.forward of the reference and oracle.ns2d.advance on identical inputs and thread counts and
prints one JSON line per case; the committed result is quoted in DESIGN.md section 7.

    PYTHONDONTWRITEBYTECODE=1 python oracle/time_vs_reference.py
"""
import json
import math
import os
import sys
import time

import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from oracle import ns2d as O  # noqa: E402


def time_loop(fn, w, seconds=6.0, min_steps=3):
    fn(w)
    t0 = time.perf_counter()
    k = 0
    while True:
        w = fn(w)
        k += 1
        el = time.perf_counter() - t0
        if (el > seconds and k >= min_steps) or k >= 400:
            return el / k, w


def case(n, B, real, forced, threads):
    torch.set_default_dtype(real)
    torch.set_num_threads(threads)
    from torch_cfd.grids import Grid
    from torch_cfd.equations import NavierStokes2DSpectral, RK4CrankNicolsonStepper
    from torch_cfd.forcings import KolmogorovForcing

    L = 2 * math.pi
    grid = Grid(shape=(n, n), domain=((0, L), (0, L)))
    forcing = KolmogorovForcing(grid=grid, scale=1.0, wave_number=4) if forced else None
    eq = NavierStokes2DSpectral(1e-3, grid, drag=0.1 if forced else 0.0, smooth=True, forcing_fn=forcing,
                                solver=RK4CrankNicolsonStepper())
    dt = 1e-3
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, real)) for s in range(B)])
    t = O.make_tables(n, L, 1e-3, 0.1 if forced else 0.0, True, None, real)
    if forced:
        t.forcing_hat = O.kolmogorov_forcing_hat(n, L, t.kx, t.ky, 1.0, 4, real=real)
    with torch.no_grad():
        ref_s, w_ref = time_loop(lambda w: eq(w, dt)[0], w0.clone())
        ora_s, w_ora = time_loop(lambda w: O.advance(w, dt, t)[0], w0.clone())
        one_ref, one_ora = eq(w0, dt)[0], O.advance(w0, dt, t)[0]
    return {"n": n, "batch": B, "dtype": str(real)[6:], "forced": forced, "threads": threads,
            "reference_ms_per_step": round(ref_s * 1e3, 2), "oracle_ms_per_step": round(ora_s * 1e3, 2),
            "oracle_over_reference": round(ora_s / ref_s, 3),
            "one_step_rel_l2": float((one_ref - one_ora).norm() / one_ref.norm())}


if __name__ == "__main__":
    thr = min(8, os.cpu_count() or 1)
    for n, B, real, forced in ((128, 1, torch.float64, True), (256, 16, torch.float32, False),
                               (512, 2, torch.float64, False), (1024, 1, torch.float64, True)):
        print(json.dumps(case(n, B, real, forced, thr)), flush=True)
