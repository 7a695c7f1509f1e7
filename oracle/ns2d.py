"""CPU oracle: batched 2-D pseudo-spectral vorticity solver (RK4 + Crank-Nicolson).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Functional restatement
in torch-CPU ops of the reference algorithm; every function cites the reference
lines it follows (paths relative to /root/reference).  Pinned against the
reference by tests/test_oracle_golden.py + tests/golden/ns2d_*.npz.

Conventions: ``w_hat`` is the half spectrum ``rfft2(w)`` of shape (*, n, m),
``m = n//2 + 1``; the box is [0, L)^2; ``real`` is the floating dtype every
constant table is built in (the reference builds them in the torch default
dtype current at construction time -- SURVEY note N1).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

# Carpenter-Kennedy 5-stage low-storage coefficients, as tabulated in
# torch_cfd/equations.py:294-317 (the only working scheme of the reference's
# RK4CrankNicolsonStepper; SURVEY headline bug 2).
CK_ALPHAS = (0.0, 0.1496590219993, 0.3704009573644, 0.6222557631345, 0.9582821306748, 1.0)
CK_BETAS = (0.0, -0.4178904745, -1.192151694643, -1.697784692471, -1.514183444257)
CK_GAMMAS = (0.1496590219993, 0.3792103129999, 0.8229550293869, 0.6994504559488, 0.1530572479681)


# --------------------------------------------------------------------------- meshes / tables
def rfft_wavenumbers(n: int, L: float, real: torch.dtype = torch.float64):
    """(kx, ky), each (n, m): ordinal frequencies of the half spectrum.

    torch_cfd/grids.py:159-201 (fft_axes -> fft_mesh -> rfft_mesh): fftfreq with
    spacing L/n on both axes, 'ij' meshgrid, last axis cut to m = n//2+1, so the
    Nyquist column holds the NEGATIVE frequency -n/2/L.
    """
    step = L / n
    ax = torch.fft.fftfreq(n, d=step, dtype=real)
    kx, ky = torch.meshgrid(ax, ax, indexing="ij")
    m = math.floor(n / 2.0) + 1
    return kx[..., :m], ky[..., :m]


def brick_wall_mask(n: int, real: torch.dtype = torch.float64) -> torch.Tensor:
    """2/3-rule mask (n, m) of zeros/ones.  torch_cfd/spectral.py:78-84
    (python-float ``int(2/3*n)`` arithmetic kept)."""
    m = n // 2 + 1
    lo_rows = int(2 / 3 * n) // 2
    # NB: the reference writes ``-int(2/3*n) // 2`` -- unary minus binds tighter
    # than //, so for odd int(2/3*n) the upper block keeps one MORE row
    # (n=128: 42 low rows, 43 high rows).
    hi_start = -int(2 / 3 * n) // 2
    keep_cols = int(2 / 3 * m)
    mask = torch.zeros((n, m), dtype=real)
    mask[:lo_rows, :keep_cols] = 1
    mask[hi_start:, :keep_cols] = 1
    return mask


def laplacian_symbol(kx, ky):
    """-4 pi^2 (|kx|^2+|ky|^2), unpatched.  torch_cfd/equations.py:398."""
    return -4 * (torch.pi) ** 2 * (abs(kx) ** 2 + abs(ky) ** 2)


def patched_laplacian(kx, ky):
    """Same symbol with the (0,0) entry set to 1 so it can be inverted.
    torch_cfd/spectral.py:41-46."""
    lap = -4 * (torch.pi**2) * (abs(kx) ** 2 + abs(ky) ** 2)
    lap[..., 0, 0] = 1
    return lap


@dataclass
class NS2DTables:
    """Constant tables of one operator instance (torch_cfd/equations.py:394-403)."""

    n: int
    L: float
    viscosity: float
    drag: float
    smooth: bool
    kx: torch.Tensor
    ky: torch.Tensor
    laplace: torch.Tensor
    linear_term: torch.Tensor
    mask: torch.Tensor
    forcing_hat: Optional[torch.Tensor] = None  # (n, m) complex, already curl'ed
    alphas: Sequence[float] = CK_ALPHAS
    betas: Sequence[float] = CK_BETAS
    gammas: Sequence[float] = CK_GAMMAS
    real: torch.dtype = torch.float64
    coef: Dict[str, torch.Tensor] = field(default_factory=dict)


def make_tables(
    n: int,
    L: float = 2 * math.pi,
    viscosity: float = 1e-3,
    drag: float = 0.0,
    smooth: bool = True,
    forcing_hat: Optional[torch.Tensor] = None,
    real: torch.dtype = torch.float64,
    alphas=CK_ALPHAS,
    betas=CK_BETAS,
    gammas=CK_GAMMAS,
) -> NS2DTables:
    """torch_cfd/equations.py:375-403 (+ RK coefficient tensors of :325-326,
    which are stored in the construction-time default dtype)."""
    kx, ky = rfft_wavenumbers(n, L, real)
    lap = laplacian_symbol(kx, ky)
    lin = viscosity * lap - drag
    # brick_wall_filter_2d builds its mask in the default dtype too
    mask = brick_wall_mask(n, real)
    t = NS2DTables(n, L, viscosity, drag, smooth, kx, ky, lap, lin, mask, forcing_hat,
                   tuple(alphas), tuple(betas), tuple(gammas), real)
    t.coef = {
        "alphas": torch.tensor(list(alphas), dtype=real),
        "betas": torch.tensor(list(betas), dtype=real),
        "gammas": torch.tensor(list(gammas), dtype=real),
    }
    return t


# --------------------------------------------------------------------------- forcings
def _axis_points(n: int, L: float, offset: float, real):
    # torch_cfd/grids.py:138-157 (Grid.axes): lower + (arange + offset) * step
    return 0.0 + (torch.arange(n, dtype=real) + offset) * (L / n)


def kolmogorov_forcing_hat(n, L, kx, ky, scale=1.0, wave_number=1, swap_xy=False,
                           vorticity=False, diam=2 * math.pi, real=torch.float64):
    """Spectral forcing term added to F in torch_cfd/equations.py:429-437 for
    torch_cfd/forcings.py:118-210 (KolmogorovForcing, offsets ((0,0),(0,0))).

    velocity form (default): curl of (fx, fy) = (scale*sin(k*y), 0) (or the
    swapped variant); vorticity form: -scale*k*cos(k*y).  State independent, so
    one (n, m) complex table.
    """
    ax = _axis_points(n, L, 0.0, real)
    X, Y = torch.meshgrid(ax, ax, indexing="ij")
    dom = 2 * torch.pi / diam
    if not vorticity:
        if swap_xy:
            fy = scale * torch.sin(wave_number * dom * X)
            fx = torch.zeros_like(fy)
        else:
            fx = scale * torch.sin(wave_number * dom * Y)
            fy = torch.zeros_like(fx)
        fx = fx.to(real)
        fy = fy.to(real)
        fxh, fyh = torch.fft.rfft2(fx), torch.fft.rfft2(fy)
        # spectral_curl_2d, torch_cfd/spectral.py:49-56
        return 2j * torch.pi * (fyh * kx - fxh * ky)
    arg = X if swap_xy else Y
    f = -scale * wave_number * dom * torch.cos(wave_number * dom * arg)
    return torch.fft.rfft2(f.to(real))


def sincos_forcing_hat(n, L, scale=0.1, k=1.0, diam=1.0, real=torch.float64):
    """torch_cfd/forcings.py:220-349 (SinCosForcing, vorticity form):
    f = s*(cos(kk(x+y)) + sin(kk(x+y))), kk = k*2pi/diam."""
    ax = _axis_points(n, L, 0.0, real)
    X, Y = torch.meshgrid(ax, ax, indexing="ij")
    kk = k * (2 * torch.pi / diam)
    f = scale * (torch.cos(kk * (X + Y)) + torch.sin(kk * (X + Y)))
    return torch.fft.rfft2(f.to(real))


# --------------------------------------------------------------------------- the operator
def stream_and_velocity(w_hat, kx, ky):
    """((u_hat, v_hat), psi_hat).  torch_cfd/spectral.py:87-115 with the
    helpers :68-75: psi = -w/lap(patched); u = 2 pi i ky psi; v = -2 pi i kx psi."""
    lap = patched_laplacian(kx, ky)
    psi = -1 / lap * w_hat
    gx = 2j * torch.pi * kx * psi
    gy = 2j * torch.pi * ky * psi
    return (gy, -gx), psi


def explicit_terms(w_hat, t: NS2DTables):
    """F(w_hat): de-aliased advection (+ forcing).  torch_cfd/equations.py:413-438."""
    (uh, vh), _ = stream_and_velocity(w_hat, t.kx, t.ky)
    vx, vy = torch.fft.irfft2(uh), torch.fft.irfft2(vh)
    dxh = 2j * torch.pi * t.kx * w_hat
    dyh = 2j * torch.pi * t.ky * w_hat
    dx, dy = torch.fft.irfft2(dxh), torch.fft.irfft2(dyh)
    adv = -(dx * vx + dy * vy)
    out = torch.fft.rfft2(adv)
    if t.smooth:
        out = out * t.mask
    if t.forcing_hat is not None:
        out = out + t.forcing_hat
    return out


def implicit_terms(w_hat, t: NS2DTables):
    """torch_cfd/equations.py:443-444."""
    return t.linear_term * w_hat


def implicit_solve(w_hat, mu, t: NS2DTables):
    """torch_cfd/equations.py:446-447."""
    return 1 / (1 - mu * t.linear_term) * w_hat


def rk4cn_step(w_hat, dt: float, t: NS2DTables):
    """One Carpenter-Kennedy RK4 / Crank-Nicolson step.  torch_cfd/equations.py:328-358."""
    al, be, ga = t.coef["alphas"], t.coef["betas"], t.coef["gammas"]
    u = w_hat
    h = 0
    for k in range(len(be)):
        h = explicit_terms(u, t) + be[k] * h
        mu = 0.5 * dt * (al[k + 1] - al[k])
        u = implicit_solve(u + ga[k] * dt * h + mu * implicit_terms(u, t), mu, t)
    return u


def advance(w_hat, dt: float, t: NS2DTables, steps: int = 1):
    """(w_new, dw/dt).  torch_cfd/equations.py:452-463."""
    old = w_hat
    for _ in range(steps):
        w_hat = rk4cn_step(w_hat, dt, t)
    return w_hat, 1 / (steps * dt) * (w_hat - old)


def residual(w_hat, wt_hat, t: NS2DTables):
    """torch_cfd/equations.py:405-411."""
    return wt_hat - explicit_terms(w_hat, t) - implicit_terms(w_hat, t)


def trajectory(w0, dt: float, t: NS2DTables, num_steps: int = 1, record_every_steps: int = 1,
               dtype: torch.dtype = torch.complex64) -> Dict[str, torch.Tensor]:
    """fno/data_gen/solvers.py:191-265 without the progress-bar residual: records
    {vorticity, stream, vort_t, residual} after every step whose 0-based index is
    a multiple of ``record_every_steps``; stacked on dim -3; cast to ``dtype``."""
    rec: Dict[str, List[torch.Tensor]] = {"vorticity": [], "stream": [], "vort_t": [], "residual": []}
    w = w0
    for step in range(num_steps):
        w, dwdt = advance(w, dt, t)
        if step % record_every_steps == 0:
            _, psi = stream_and_velocity(w, t.kx, t.ky)
            res = residual(w, dwdt, t)
            for key, val in zip(("vorticity", "vort_t", "stream", "residual"), (w, dwdt, psi, res)):
                rec[key].append(val.to(dtype).clone())
    return {k: torch.stack(v, dim=-3) for k, v in rec.items()}


def stable_time_step(dx, dt=None, max_velocity=1.0, max_courant_number=0.5, viscosity=1e-3,
                     implicit_diffusion=True, ndim=2):
    """torch_cfd/equations.py:35-64."""
    dt_diff = dx if implicit_diffusion else dx**2 / (viscosity * 2**ndim)
    dt_adv = max_courant_number * dx / max_velocity
    dt = dt_adv if dt is None else dt
    return min(dt_diff, dt_adv, dt)


# --------------------------------------------------------------------------- initial condition
def mcwilliams_vorticity(n: int, L: float, peak_wavenumber: float = 3.0, seed: int = 0,
                         real: torch.dtype = torch.float64) -> torch.Tensor:
    """McWilliams random vorticity, physical space (n, n).

    torch_cfd/initial_conditions.py:170-199 with :68-107: white noise from a
    seeded CPU generator -> filter with k^-1 (1 + (k/k0)^4)^-1 -> normalise the
    stream function to unit kinetic energy -> vorticity = ifft(k^2 fft(psi)).
    ``real`` plays the role of the default dtype (the randn stream differs
    between float32 and float64, SURVEY 8c work-arounds).
    """
    gen = torch.Generator()
    gen.manual_seed(seed)
    noise = torch.randn((n, n), generator=gen, dtype=real)
    om = 2 * torch.pi * torch.fft.fftfreq(n, L / n, dtype=real)
    kvec = torch.stack(torch.meshgrid(om, om, indexing="ij"), dim=0)
    k = torch.linalg.norm(kvec, dim=0)
    dens = (k * (1.0 + (k / peak_wavenumber) ** 4)) ** (-1)
    filt = torch.where(k > 0, dens, 0.0)
    psi = torch.fft.ifftn(torch.fft.fftn(noise) * filt).real
    psih = torch.fft.fft2(psi)
    ke = (2 * (k * psih).abs() ** 2 / (n * n) ** 2).sum()
    psi = psi / ke.sqrt()
    return torch.fft.ifftn(torch.fft.fftn(psi) * k**2).real


def filtered_velocity_field(n: int, L: float, maximum_velocity: float = 1.0, peak_wavenumber: float = 3.0,
                            iterations: int = 3, seed: int = 0, real: torch.dtype = torch.float64):
    """Divergence-free random velocity on the staggered (MAC) grid, components (ux, uy) at the cell faces.

    torch_cfd/initial_conditions.py:122-167: per component white noise (seeds ``seed``, ``seed+1``) filtered with a
    log-normal density / k (:60-66, :89-99), then ``iterations`` x project_and_normalize (:110-119):
    backward-difference divergence (finite_differences.py:126-135) -> pseudo-inverse of the FINITE-DIFFERENCE
    Laplacian by circulant rfftn diagonalisation (pressure.py:296-360: eigenvalues = fft / rfft of the first column
    [-2, 1, 0.., 1]/h^2 of the periodic 1-D Laplacian, finite_differences.py:167-193; inverse where |lambda| >
    10 eps(float32), the solver's default dtype) -> subtract the forward-difference gradient (:74-83) -> rescale
    so that max |u| = maximum_velocity.  ``real`` plays the role of the default dtype.
    """
    h = L / n
    om = 2 * torch.pi * torch.fft.fftfreq(n, h, dtype=real)
    k = torch.linalg.norm(torch.stack(torch.meshgrid(om, om, indexing="ij"), dim=0), dim=0)
    variance = 0.25
    mean = math.log(peak_wavenumber) + variance
    logk = torch.log(k)
    dens = torch.exp(-((mean - logk) ** 2) / 2 / variance - logk) / k
    filt = torch.where(k > 0, dens, 0.0)
    comps = []
    gen = torch.Generator()
    for i in range(2):
        gen.manual_seed(seed + i)
        noise = torch.randn((n, n), generator=gen, dtype=real)
        comps.append(torch.fft.ifftn(torch.fft.fftn(noise) * filt).real)
    ux, uy = comps
    col = torch.zeros(n, dtype=real)
    col[0] = -2 / h**2
    col[1] = col[-1] = 1 / h**2
    lam = torch.fft.fft(col)[:, None] + torch.fft.rfft(col)[None, :]
    cutoff = 10 * torch.finfo(torch.float32).eps
    inv = torch.where(torch.abs(lam) > cutoff, 1 / lam, 0)
    for _ in range(iterations):
        div = (ux - torch.roll(ux, 1, 0)) / h + (uy - torch.roll(uy, 1, 1)) / h
        q = torch.fft.irfftn(inv * torch.fft.rfftn(div), s=(n, n)).real
        ux = ux - (torch.roll(q, -1, 0) - q) / h
        uy = uy - (torch.roll(q, -1, 1) - q) / h
        vmax = torch.linalg.norm(torch.stack([ux, uy]), dim=0).max()
        ux, uy = maximum_velocity * ux / vmax, maximum_velocity * uy / vmax
    return ux, uy


def curl_2d(ux: torch.Tensor, uy: torch.Tensor, L: float) -> torch.Tensor:
    """Forward-difference curl of a staggered velocity, torch_cfd/finite_differences.py:412-419."""
    h = L / ux.shape[-1]
    return (torch.roll(uy, -1, -2) - uy) / h - (torch.roll(ux, -1, -1) - ux) / h
