"""Host-side pieces of the k-space arithmetic that callers of the reference name directly.

Only two functions of torch_cfd/spectral.py are part of the operator API the
spectral path is used through: ``brick_wall_filter_2d`` (the 2/3-rule mask that
becomes a plan table, spectral.py:78-84) and ``vorticity_to_velocity``
(spectral.py:87-115, here one launch of ``k_velocity``).  Everything else the
reference composes per RK stage from small helpers (Laplacian, gradient, rot,
curl) is arithmetic inside the HIP kernels (csrc/tcfd_ns2d.hip: ``emit_planes``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .grids import Grid


def dealias_extents(n: int) -> Tuple[int, int, int]:
    """(low rows kept, high rows kept, columns kept) of the 2/3-rule brick wall on an n x (n//2+1) half spectrum.

    The reference builds the mask with Python slices ``[: int(2/3*n)//2]`` and ``[-int(2/3*n)//2 :]``
    (spectral.py:82-83).  Unary minus binds tighter than ``//``, so the upper block keeps
    ``ceil(k/2)`` rows while the lower one keeps ``floor(k/2)``, k = int(2/3*n): n = 128 keeps 42 low and
    43 high rows.  The golden tables (tests/golden/ns2d_tables.npz) pin this.
    """
    k = int(2 / 3 * n)
    cols = int(2 / 3 * (n // 2 + 1))
    return k // 2, -(-k // 2), cols


def brick_wall_filter_2d(grid: Grid) -> torch.Tensor:
    """0/1 mask (n, n//2+1) in the default dtype on the CPU, as the reference returns it."""
    n = grid.shape[0]
    low, high, cols = dealias_extents(n)
    i = torch.arange(n)
    j = torch.arange(n // 2 + 1)
    rows_kept = (i < low) | (i >= n - high)
    return (rows_kept[:, None] & (j < cols)[None, :]).to(torch.get_default_dtype())


def vorticity_to_velocity(grid: Grid, w_hat: torch.Tensor,
                          rfft_mesh: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """``((u_hat, v_hat), psi_hat)`` of a vorticity half spectrum ``(*, n, m)`` on a HIP device:
    psi = -w / lap (lap(0,0) := 1), u = 2 pi i ky psi, v = -2 pi i kx psi -- one launch of ``k_velocity``."""
    from .equations import _plan_for_mesh  # late: equations imports this module

    kx, ky = grid.rfft_mesh() if rfft_mesh is None else rfft_mesh
    if tuple(kx.shape[-2:]) != tuple(w_hat.shape[-2:]):
        raise ValueError(f"wavenumber mesh {tuple(kx.shape[-2:])} does not match the spectrum {tuple(w_hat.shape[-2:])}")
    return _plan_for_mesh(kx, ky, w_hat).velocity(w_hat)


# ----------------------------------------------------------------------------- k-space helpers callers name directly
# Plain tensor expressions on whatever device their arguments live on (torch_cfd/spectral.py:29-75).  The operator
# itself never calls them -- the same arithmetic is fused into the HIP kernels -- they are here so that scripts written
# against the reference (diagnostics, custom forcings, post-processing) keep working.
_TWO_PI_I = 2j * torch.pi


def fft_mesh_2d(n: int, diam: float, device=None):
    """Full (n, n) wavenumber mesh (cycles per unit length), ``indexing='ij'``."""
    k = torch.fft.fftfreq(n, d=diam / n)
    kx, ky = torch.meshgrid(k, k, indexing="ij")
    return kx.to(device), ky.to(device)


def spectral_laplacian_2d(fft_mesh, device=None) -> torch.Tensor:
    """Symbol of the Laplacian, ``-(2 pi)^2 |k|^2``, with the (0, 0) entry set to 1 so that it can be divided by."""
    kx, ky = fft_mesh
    lap = -((2 * torch.pi) ** 2) * (kx.abs() ** 2 + ky.abs() ** 2)
    lap[..., 0, 0] = 1
    return lap.to(device)


def spectral_grad_2d(f_hat: torch.Tensor, rfft_mesh):
    """(d/dx, d/dy) of a scalar half spectrum."""
    kx, ky = rfft_mesh
    return _TWO_PI_I * kx * f_hat, _TWO_PI_I * ky * f_hat


def spectral_rot_2d(psi_hat: torch.Tensor, rfft_mesh):
    """Velocity of a stream function: (d psi / dy, -d psi / dx)."""
    dx, dy = spectral_grad_2d(psi_hat, rfft_mesh)
    return dy, -dx


def spectral_curl_2d(vel_hat, rfft_mesh) -> torch.Tensor:
    """Scalar curl dv/dx - du/dy of a velocity pair of half spectra."""
    u_hat, v_hat = vel_hat
    kx, ky = rfft_mesh
    return _TWO_PI_I * (kx * v_hat - ky * u_hat)


def spectral_div_2d(vel_hat, rfft_mesh) -> torch.Tensor:
    """Divergence du/dx + dv/dy of a velocity pair of half spectra."""
    u_hat, v_hat = vel_hat
    kx, ky = rfft_mesh
    return _TWO_PI_I * (kx * u_hat + ky * v_hat)
