"""k-space helper tables and the vorticity -> velocity map.

Host-side table builders keep the reference names (torch_cfd/spectral.py:29-84)
because callers construct operators with them; the per-step arithmetic the
reference does with these helpers (spectral.py:41-75, 87-115) lives in the HIP
kernels (csrc/tcfd_ns2d.hip: ``emit_planes``, ``k_velocity``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .grids import Grid


def fft_mesh_2d(n, diam, device=None):
    """Full (n, n) ordinal wavenumber mesh (spectral.py:29-32)."""
    k = torch.fft.fftfreq(n, d=diam / n)
    kx, ky = torch.meshgrid([k, k], indexing="ij")
    return kx.to(device), ky.to(device)


def fft_expand_dims(fft_mesh, batch_size):
    """(x, y) -> (b, x, y, 1) broadcast copies (spectral.py:35-38)."""
    kx, ky = fft_mesh
    return tuple(z[None, :, :, None].expand(batch_size, -1, -1, 1) for z in (kx, ky))


def spectral_laplacian_2d(fft_mesh, device=None):
    """-4 pi^2 |k|^2 with the (0, 0) entry set to 1 so it can be inverted (spectral.py:41-46)."""
    kx, ky = fft_mesh
    lap = -4 * (torch.pi**2) * (abs(kx) ** 2 + abs(ky) ** 2)
    lap[..., 0, 0] = 1
    return lap.to(device)


def spectral_curl_2d(vhat, rfft_mesh):
    uhat, vhat = vhat
    kx, ky = rfft_mesh
    return 2j * torch.pi * (vhat * kx - uhat * ky)


def spectral_div_2d(vhat, rfft_mesh):
    uhat, vhat = vhat
    kx, ky = rfft_mesh
    return 2j * torch.pi * (uhat * kx + vhat * ky)


def spectral_grad_2d(vhat, rfft_mesh):
    kx, ky = rfft_mesh
    return 2j * torch.pi * kx * vhat, 2j * torch.pi * ky * vhat


def spectral_rot_2d(vhat, rfft_mesh):
    gx, gy = spectral_grad_2d(vhat, rfft_mesh)
    return gy, -gx


def brick_wall_filter_2d(grid: Grid):
    """2/3-rule mask (n, n//2+1).  Same arithmetic as spectral.py:78-84,
    including its ``-int(2/3*n) // 2`` precedence (the upper row block keeps
    ceil(k/2) rows, the lower one floor(k/2))."""
    n, _ = grid.shape
    m = n // 2 + 1
    mask = torch.zeros((n, m))
    cols = int(2 / 3 * m)
    mask[: int(2 / 3 * n) // 2, :cols] = 1
    mask[-int(2 / 3 * n) // 2:, :cols] = 1
    return mask


def vorticity_to_velocity(grid: Grid, w_hat: torch.Tensor,
                          rfft_mesh: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """((u_hat, v_hat), psi_hat) from the vorticity half spectrum (spectral.py:87-115).

    Runs the HIP kernel ``k_velocity`` on the tensor's device; ``w_hat`` must be
    a complex64/complex128 HIP tensor (*, n, m).
    """
    from .equations import _plan_for_mesh  # late import: equations imports this module

    kx, ky = rfft_mesh if rfft_mesh is not None else grid.rfft_mesh()
    assert kx.shape[-2:] == w_hat.shape[-2:]
    plan = _plan_for_mesh(kx, ky, w_hat)
    return plan.velocity(w_hat)
