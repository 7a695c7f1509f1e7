"""CPU oracle: mode-truncated spectral convolution of the FNO / SFNO layer.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Functional restatement
in torch-CPU ops; pinned against reference outputs by
tests/test_oracle_golden_fno.py + tests/golden/fno_*.npz.

Weights are passed as a list of 4 complex tensors (Ci, Co, mx, my, mt) in the
reference's block order ``ix + 2*iy`` = [lo-x lo-y, hi-x lo-y, lo-x hi-y,
hi-x hi-y] (fno/sfno.py:374-391; fno/fno3d.py:101-112 uses weights1..4 in the
same order).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F


def corner_blocks(mx: int, my: int):
    sx = [slice(0, mx), slice(-mx, None)]
    sy = [slice(0, my), slice(-my, None)]
    return [(sx[ix], sy[iy]) for iy in range(2) for ix in range(2)]  # index = ix + 2*iy


def spectral_contract(vh: torch.Tensor, weights: Sequence[torch.Tensor], modes, bias=None, delta: float = 1.0):
    """out[b, o, X-block, Y-block, :mt] = sum_i vh[b, i, block] * W[i, o, block] (+ delta * bias[block]);
    zero elsewhere.  fno/sfno.py:364-391, fno/base.py:189, fno/fno3d.py:83-112."""
    mx, my, mt = modes
    b, ci, kx, ky, kt = vh.shape
    co = weights[0].shape[1]
    out = torch.zeros(b, co, kx, ky, kt, dtype=vh.dtype)
    for idx, (sx, sy) in enumerate(corner_blocks(mx, my)):
        blk = torch.einsum("bixyt,ioxyt->boxyt", vh[:, :, sx, sy, :mt], weights[idx])
        out[:, :, sx, sy, :mt] = blk
        if bias is not None:
            out[:, :, sx, sy, :mt] += delta * bias[idx][None, None]
    return out


def spectral_conv(v: torch.Tensor, weights, modes, bias=None, delta: float = 1.0, norm: str = "backward",
                  out_size=None):
    """rfftn over (x, y, t) -> contraction -> irfftn(s=out_size).  fno/base.py:229-237."""
    vh = torch.fft.rfftn(v, dim=(-3, -2, -1), norm=norm)
    oh = spectral_contract(vh, weights, modes, bias, delta)
    s = tuple(v.shape[-3:]) if out_size is None else tuple(out_size)
    return torch.fft.irfftn(oh, s=s, dim=(-3, -2, -1), norm=norm)


def spectral_conv_t(v: torch.Tensor, weights, modes, bias=None, delta: float = 0.1, out_steps: Optional[int] = None,
                    temporal_padding: bool = False, norm: str = "backward"):
    """Time-resampling variant: optional left zero pad by T, contraction, inverse
    transform to out_steps (+ pad), keep the last out_steps.  fno/sfno.py:433-457."""
    t_pad = v.size(-1) if temporal_padding else 0
    if temporal_padding:
        v = F.pad(v, (t_pad, 0))
    nx, ny, ntp = v.shape[-3:]
    vh = torch.fft.rfftn(v, dim=(-3, -2, -1), norm=norm)
    oh = spectral_contract(vh, weights, modes, bias, delta)
    out = torch.fft.irfftn(oh, s=(nx, ny, out_steps + t_pad), dim=(-3, -2, -1), norm=norm)
    return out[..., -out_steps:] if temporal_padding else out


def sobolev_loss(x: torch.Tensor, y: torch.Tensor, n_grid: int, norm_order: float = -1, alpha: float = 0.1,
                 diam: float = 1.0, relative: bool = False, time_average: bool = True, reduction: bool = True,
                 mesh_weighted: bool = True, fft_norm: str = "backward"):
    """fno/losses.py:263-315 (freq_cutoff=None).  NB norm_order == 0 still multiplies
    by sqrt(alpha + 4 pi^2 |k|^2) (the code, not its comment -- SURVEY a18)."""
    n = n_grid
    k = torch.fft.fftfreq(n, d=diam / n)
    kx, ky = torch.meshgrid(k, k, indexing="ij")
    cutoff = (n // 2 + 1) / diam
    fill = math.inf if norm_order < 0 else 0.0
    kx = kx.clone().masked_fill(kx.abs() > cutoff, fill)[None, :, :, None]
    ky = ky.clone().masked_fill(ky.abs() > cutoff, fill)[None, :, :, None]
    weight = torch.sqrt(alpha + 4 * math.pi**2 * (kx**2 + ky**2))
    w = weight ** (norm_order / 2) if norm_order != 0 else weight
    bsz, nt = x.shape[0], x.shape[-1]
    xh = torch.fft.fftn(x, dim=(1, 2), norm=fft_norm) * w
    yh = (torch.fft.fftn(y, dim=(1, 2), norm=fft_norm) if y is not None else torch.zeros_like(xh)) * w   # losses.py:283-286
    diff = torch.linalg.norm(xh - yh, dim=(1, 2))
    if relative:
        yn = (torch.linalg.norm(yh, dim=(1, 2)) ** 2).sum(dim=-1).sqrt()
    else:
        yn = torch.ones(bsz)
    loss = (diff**2).sum(dim=-1).sqrt()
    yn = yn / n if mesh_weighted else yn
    loss = loss / yn
    loss = loss / math.sqrt(nt) if time_average else loss
    loss = loss.mean(0) if reduction else loss.sum(0)
    return loss / n if mesh_weighted else loss
