"""rfftn / irfftn over the last ``dim`` axes as dense DFT matrix products on the device.

The hand-written FNO kernels are the (2+1)-D layer of the reference (``SpectralConvS`` and its subclasses).  The reference's
layer TEMPLATE, ``fno/base.py:114-237``, is dimension-generic: ``forward`` = rfftn over the last ``dim`` axes, the
subclass's ``spectral_conv``, irfftn to ``out_mesh_size``.  To keep that template usable for dim = 1, 2, 4, ... without
torch.fft on the data path, the two transforms are written here as one matrix product per axis (rocBLAS on the device):
O(N n) per axis instead of O(N log n), any axis length, differentiable through autograd like any other matmul.  torch's
semantics are reproduced exactly: the spectrum array is trimmed / zero-padded at its END for another output size, and the
c2r step along the last axis ignores the imaginary parts of its DC and Nyquist entries.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import torch

_TABLES: Dict[Tuple, torch.Tensor] = {}


def _norm_factor(norm, n_total: int, inverse: bool) -> float:
    if norm in (None, "backward"):
        return 1.0 / n_total if inverse else 1.0
    if norm == "ortho":
        return 1.0 / math.sqrt(n_total)
    if norm == "forward":
        return 1.0 if inverse else 1.0 / n_total
    raise ValueError(f"unknown fft norm {norm!r}")


def _angles(rows: int, cols: int, n: int, device) -> torch.Tensor:
    """2 pi (j k mod n) / n for j < rows, k < cols, reduced in integer arithmetic."""
    j = torch.arange(rows, device=device)
    k = torch.arange(cols, device=device)
    return (2 * math.pi / n) * ((j[:, None] * k[None, :]) % n).to(torch.float64)


def _table(kind: str, n: int, cols: int, real: torch.dtype, device) -> torch.Tensor:
    key = (kind, n, cols, real, str(device))
    t = _TABLES.get(key)
    if t is None:
        cdt = torch.complex128 if real == torch.float64 else torch.complex64
        if kind in ("fwd", "inv"):          # (n_in = cols, n) complex: x[..., j] -> X[..., k] = sum_j x_j e^{-+ 2 pi i j k / n}
            ang = _angles(cols, n, n, device)
            t = torch.polar(torch.ones_like(ang), -ang if kind == "fwd" else ang).to(cdt)
        elif kind in ("r2c_re", "r2c_im"):  # (n, m) real: the half transform of a real axis
            ang = _angles(n, cols, n, device)
            t = (torch.cos(ang) if kind == "r2c_re" else -torch.sin(ang)).to(real)
        else:                               # "c2r_re" / "c2r_im": (cols = kept columns, n) real, weights 1, 2, ..., 2[, 1]
            ang = _angles(cols, n, n, device)
            c = torch.full((cols,), 2.0, dtype=torch.float64, device=device)
            c[0] = 1.0
            t = c[:, None] * (torch.cos(ang) if kind == "c2r_re" else torch.sin(ang))
            if n % 2 == 0 and cols > n // 2:
                t[n // 2] *= 0.5
                if kind == "c2r_im":
                    t[n // 2].zero_()
            if kind == "c2r_im":
                t[0].zero_()
            t = t.to(real)
        _TABLES[key] = t
    return t


def _along(x: torch.Tensor, axis: int, mat: torch.Tensor) -> torch.Tensor:
    """Contract axis ``axis`` of x with the rows of ``mat`` (n_in, n_out)."""
    return torch.movedim(torch.movedim(x, axis, -1) @ mat, -1, axis)


def rfftn_dense(v: torch.Tensor, dim: int, norm="backward") -> torch.Tensor:
    """``torch.fft.rfftn(v, dim=last dim axes, norm=norm)`` as matrix products."""
    if v.is_complex() or v.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"expected a real fp32 / fp64 tensor, got {v.dtype}")
    sizes = v.shape[-dim:]
    n = sizes[-1]
    m = n // 2 + 1
    h = torch.complex(v @ _table("r2c_re", n, m, v.dtype, v.device), v @ _table("r2c_im", n, m, v.dtype, v.device))
    for a in range(dim - 1):
        ax = -dim + a
        h = _along(h, ax, _table("fwd", sizes[a], sizes[a], v.dtype, v.device))
    f = _norm_factor(norm, math.prod(sizes), inverse=False)
    return h * f if f != 1.0 else h


def irfftn_dense(vh: torch.Tensor, s: Sequence[int], norm="backward") -> torch.Tensor:
    """``torch.fft.irfftn(vh, s=s, dim=last len(s) axes, norm=norm)`` as matrix products: every axis of the spectrum array is
    trimmed or zero-padded at its end to the output size first (what torch does), the last axis to s[-1] // 2 + 1 columns."""
    if not vh.is_complex():
        raise TypeError("expected a complex spectrum")
    dim = len(s)
    real = torch.float64 if vh.dtype == torch.complex128 else torch.float32
    g = vh
    for a in range(dim - 1):
        ax = -dim + a
        keep = min(g.shape[ax], s[a])
        g = _along(g.narrow(ax, 0, keep), ax, _table("inv", s[a], keep, real, vh.device))
    n = s[-1]
    keep = min(g.shape[-1], n // 2 + 1)
    g = g[..., :keep]
    y = g.real @ _table("c2r_re", n, keep, real, vh.device) - g.imag @ _table("c2r_im", n, keep, real, vh.device)
    f = _norm_factor(norm, math.prod(s), inverse=True)
    return y * f if f != 1.0 else y
