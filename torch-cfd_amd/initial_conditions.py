"""Random initial vorticity with the McWilliams (1984) spectrum, generated on the device.

Drop-in for ``vorticity_field`` of the reference (torch_cfd/initial_conditions.py:170-199
with ``spectral_filter`` :89-99, ``streamfunc_normalize`` :102-107,
``McWilliams_density`` :68-77).  The white noise comes from the same seeded CPU
generator (so a given ``random_state`` reproduces the reference's sample), then
everything runs through the HIP rfft2/irfft2 kernels.  The reference's complex
fftn/ifftn of a real field times a real even filter is exactly an rfft2/irfft2
pair, which is what is used here.
"""
from __future__ import annotations

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


def vorticity_field(grid: Grid, peak_wavenumber: float = 3, random_state: int = 0, device="cuda",
                    batch_seeds=None) -> torch.Tensor:
    """(n, n) real vorticity on ``device`` (or (B, n, n) when ``batch_seeds`` lists
    several seeds).  dtype follows the torch default dtype, like the reference."""
    real = torch.get_default_dtype()
    n = grid.shape[0]
    seeds = [random_state] if batch_seeds is None else list(batch_seeds)
    gen = torch.Generator()
    noise = []
    for s in seeds:
        gen.manual_seed(int(s))
        noise.append(torch.randn(grid.shape, generator=gen))
    noise = torch.stack(noise).to(device)
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
