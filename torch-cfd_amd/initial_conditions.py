"""Random initial conditions generated on the device: McWilliams (1984) vorticity and the filtered,
divergence-free staggered velocity field (+ its finite-difference curl) of BASELINE config 1.

Drop-in for ``vorticity_field`` of the reference (torch_cfd/initial_conditions.py:170-199
with ``spectral_filter`` :89-99, ``streamfunc_normalize`` :102-107,
``McWilliams_density`` :68-77).  The white noise comes from the same seeded CPU
generator (so a given ``random_state`` reproduces the reference's sample), then
everything runs through the HIP rfft2/irfft2 kernels.  The reference's complex
fftn/ifftn of a real field times a real even filter is exactly an rfft2/irfft2
pair, which is what is used here.
"""
from __future__ import annotations

import math

import torch

from .equations import _COMPLEX_OF, fft_plan
from .grids import Grid


def McWilliams_density(k, mode: float, tau: float = 1.0):
    """|psi|^2 ~ k^-1 (tau^2 + (k/k0)^4)^-1."""
    return (k * (tau**2 + (k / mode) ** 4)) ** (-1)


def _half_angular_magnitude(grid: Grid, device, real):
    n = grid.shape[0]
    om = [2 * torch.pi * torch.fft.fftfreq(s, h).to(real) for s, h in zip(grid.shape, grid.step)]
    kx, ky = torch.meshgrid(*om, indexing="ij")
    k = torch.sqrt(kx**2 + ky**2)[:, : n // 2 + 1]
    return k.to(device)


def _seeded_noise(shape, seeds) -> torch.Tensor:
    """(len(seeds), *shape) standard-normal samples, sample i from a CPU generator seeded with seeds[i] -- the reference's
    per-sample stream (its generators live on the CPU), so a seed reproduces the reference's field.  The generators are
    independent: large batches are drawn by a few threads (the draw releases the GIL) straight into one buffer."""
    out = torch.empty((len(seeds),) + tuple(shape))

    def draw(i):
        gen = torch.Generator()
        gen.manual_seed(int(seeds[i]))
        torch.randn(tuple(shape), generator=gen, out=out[i])

    if len(seeds) < 4:
        for i in range(len(seeds)):
            draw(i)
    else:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=min(16, len(seeds))) as pool:
            list(pool.map(draw, range(len(seeds))))
    return out


def vorticity_field(grid: Grid, peak_wavenumber: float = 3, random_state: int = 0, device="cuda",
                    batch_seeds=None) -> torch.Tensor:
    """(n, n) real vorticity on ``device`` (or (B, n, n) when ``batch_seeds`` lists
    several seeds).  dtype follows the torch default dtype, like the reference."""
    real = torch.get_default_dtype()
    n = grid.shape[0]
    seeds = [random_state] if batch_seeds is None else list(batch_seeds)
    noise = _seeded_noise(grid.shape, seeds).to(device)
    plan = fft_plan(n, _COMPLEX_OF[real], torch.device(device), diam=grid.domain[0][1] - grid.domain[0][0])
    k = _half_angular_magnitude(grid, device, real)
    filt = torch.where(k > 0, McWilliams_density(k, peak_wavenumber), torch.zeros_like(k))
    psi_hat = plan.rfft2(noise) * filt
    # kinetic-energy normalisation: sum over the FULL spectrum of 2 |k psi^|^2 / n^4
    weight = torch.full((n // 2 + 1,), 2.0, dtype=real, device=device)
    weight[0] = 1.0
    weight[-1] = 1.0
    ke = ((k * psi_hat).abs() ** 2 * weight).sum(dim=(-2, -1)) * 2 / (n * n) ** 2
    psi_hat = psi_hat / ke.sqrt()[:, None, None]
    w = plan.irfft2(psi_hat * k**2)
    return w[0] if batch_seeds is None else w


def _log_normal_density(k, mode: float, variance: float = 0.25):
    """Unscaled log-normal pdf peaked at ``mode`` (torch_cfd/initial_conditions.py:60-66)."""
    mean = math.log(mode) + variance
    logk = torch.log(k)
    return torch.exp(-((mean - logk) ** 2) / 2 / variance - logk)


def filtered_velocity_field(grid: Grid, maximum_velocity: float = 1, peak_wavenumber: float = 3, iterations: int = 3,
                            random_state: int = 0, device="cuda", batch_seeds=None):
    """Divergence-free random velocity ``(ux, uy)`` on the staggered grid (ux at the x-faces, uy at the y-faces),
    each (n, n) on ``device`` (or (B, n, n) when ``batch_seeds`` lists several seeds).

    Drop-in for ``filtered_velocity_field`` of the reference (torch_cfd/initial_conditions.py:122-167), which runs on
    the CPU through the finite-volume stack: per component white noise from the same seeded CPU generator (seeds
    ``random_state``, ``random_state + 1``), log-normal spectral filter, then ``iterations`` sweeps of
    {backward-difference divergence, pseudo-inverse of the FINITE-DIFFERENCE Laplacian (circulant eigenvalues =
    fft / rfft of its first column, cut-off 10 eps(float32): pressure.py:296-360), subtract the forward-difference
    gradient, rescale to ``maximum_velocity``}.  The transforms are the HIP rfft2 / irfft2 kernels; the stencils and
    the rescaling are element-wise device ops.  Returns plain tensors (the reference wraps them in a
    GridVariableVector, which is outside this path).
    """
    real = torch.get_default_dtype()
    n = grid.shape[0]
    h = grid.step[0]
    seeds = [random_state] if batch_seeds is None else list(batch_seeds)
    noise = _seeded_noise(grid.shape, [int(s) + i for s in seeds for i in range(2)]).to(device)   # (2B, n, n): ux, uy interleaved
    plan = fft_plan(n, _COMPLEX_OF[real], torch.device(device), diam=grid.domain[0][1] - grid.domain[0][0])
    k = _half_angular_magnitude(grid, device, real)
    kk = torch.where(k > 0, k, torch.ones_like(k))
    filt = torch.where(k > 0, _log_normal_density(kk, peak_wavenumber) / kk, torch.zeros_like(k))
    v = plan.irfft2(plan.rfft2(noise) * filt)
    ux, uy = v[0::2].contiguous(), v[1::2].contiguous()
    col = torch.zeros(n, dtype=real)
    col[0] = -2 / h**2
    col[1] = col[-1] = 1 / h**2
    lam = torch.fft.fft(col)[:, None] + torch.fft.rfft(col)[None, :]
    inv = torch.where(torch.abs(lam) > 10 * torch.finfo(torch.float32).eps, 1 / lam, 0).to(device)
    for _ in range(iterations):
        div = (ux - torch.roll(ux, 1, -2)) / h + (uy - torch.roll(uy, 1, -1)) / h
        q = plan.irfft2(plan.rfft2(div) * inv)
        ux = ux - (torch.roll(q, -1, -2) - q) / h
        uy = uy - (torch.roll(q, -1, -1) - q) / h
        vmax = torch.sqrt(ux * ux + uy * uy).amax(dim=(-2, -1), keepdim=True)
        ux, uy = maximum_velocity * ux / vmax, maximum_velocity * uy / vmax
    if batch_seeds is None:
        return ux[0], uy[0]
    return ux, uy


def curl_2d(v, grid: Grid) -> torch.Tensor:
    """Forward-difference curl of a staggered velocity ``(ux, uy)`` (torch_cfd/finite_differences.py:412-419)."""
    ux, uy = v
    hx, hy = grid.step
    return (torch.roll(uy, -1, -2) - uy) / hx - (torch.roll(ux, -1, -1) - ux) / hy
