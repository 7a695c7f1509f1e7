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


# ----------------------------------------------------------------------------- truncated (2+1)-D transforms, any X / Y
# The fused FNO kernels (csrc/tcfd_fno.hip) transform power-of-two X and Y.  Every other spatial size -- a 96^2 / 192^2
# data set, the X + 2p grid of ``SFNO(spatial_padding=p)`` (fno/sfno.py:313-328) -- runs the SAME pruned transforms as
# matrix products with the kept rows of the DFT matrices: only 2mx x 2my x mt modes are ever formed
# (fno/sfno.py:379-381), so the matrices are (X, 2mx), (Y, 2my), (T, mt) -- three thin GEMMs per direction (rocBLAS),
# differentiable through autograd.  torch's semantics as above: the left zero padding in t is skipped arithmetically, an
# output grid other than the input's keeps the high-frequency block at the ARRAY indices it had on the source grid.
def _kept_indices(n: int, m: int, device) -> torch.Tensor:
    """Array indices of the rows [:m] + [-m:] of a length-n spectrum axis, in truncated order."""
    return torch.cat([torch.arange(m, device=device), torch.arange(n - m, n, device=device)])


def _trunc_table(kind: str, key: Tuple, real: torch.dtype, device) -> torch.Tensor:
    full = ("trunc", kind) + key + (real, str(device))
    t = _TABLES.get(full)
    if t is not None:
        return t
    cdt = torch.complex128 if real == torch.float64 else torch.complex64
    if kind == "fwd_xy":            # (n, 2m): e^{-2 pi i j k / n} at the kept k
        n, m = key
        ang = (2 * math.pi / n) * ((torch.arange(n, device=device)[:, None] * _kept_indices(n, m, device)[None, :]) % n).to(torch.float64)
        t = torch.polar(torch.ones_like(ang), -ang).to(cdt)
    elif kind == "fwd_t":           # (T, mt) complex: e^{-2 pi i kt (t + t_pad) / Tp}
        T, t_pad, mt = key
        Tp = T + t_pad
        tt = torch.arange(T, device=device) + t_pad
        ang = (2 * math.pi / Tp) * ((tt[:, None] * torch.arange(mt, device=device)[None, :]) % Tp).to(torch.float64)
        t = torch.polar(torch.ones_like(ang), -ang).to(cdt)
    elif kind == "inv_xy":          # (2m, n_out): e^{+2 pi i j a / n_out}, a = array index on the SOURCE grid; rows beyond n_out vanish
        n_out, n_src, m = key
        a = _kept_indices(n_src, m, device)
        ang = (2 * math.pi / n_out) * ((a[:, None] * torch.arange(n_out, device=device)[None, :]) % n_out).to(torch.float64)
        t = torch.polar(torch.ones_like(ang), ang)
        t = torch.where((a < n_out)[:, None], t, torch.zeros_like(t)).to(cdt)
    else:                           # "inv_t_re" / "inv_t_im": (mt, t_keep) real c2r rows of the LAST t_keep output steps
        T_out, t_keep, mt = key
        k = torch.arange(mt, device=device)
        tt = torch.arange(T_out - t_keep, T_out, device=device)
        ang = (2 * math.pi / T_out) * ((k[:, None] * tt[None, :]) % T_out).to(torch.float64)
        c = torch.full((mt,), 2.0, dtype=torch.float64, device=device)
        c[0] = 1.0
        if T_out % 2 == 0 and mt > T_out // 2:
            c[T_out // 2] = 1.0
        c = torch.where(k <= T_out // 2, c, torch.zeros_like(c))        # torch trims the spectrum to T_out // 2 + 1 columns
        t = (c[:, None] * (torch.cos(ang) if kind == "inv_t_re" else torch.sin(ang))).to(real)
    _TABLES[full] = t
    return t


def truncated_rfftn_dense(v: torch.Tensor, modes: Sequence[int], t_pad: int = 0) -> torch.Tensor:
    """Kept modes of ``rfftn(left_pad_t(v, t_pad))`` (unnormalised): (b, C, X, Y, T) real -> (b, C, 2mx, 2my, mt) complex."""
    b, c, X, Y, T = v.shape
    mx, my, mt = modes
    if 2 * mx > X or 2 * my > Y or mt > (T + t_pad) // 2 + 1:
        raise ValueError(f"modes {tuple(modes)} exceed the spectrum of a ({X}, {Y}, {T + t_pad}) grid")
    real, dev = v.dtype, v.device
    at = _trunc_table("fwd_t", (T, t_pad, mt), real, dev)
    h = torch.complex(v @ at.real.contiguous(), v @ at.imag.contiguous())          # (b, C, X, Y, mt)
    h = torch.einsum("bcxyt,yk->bcxkt", h, _trunc_table("fwd_xy", (Y, my), real, dev))
    return torch.einsum("bcxkt,xj->bcjkt", h, _trunc_table("fwd_xy", (X, mx), real, dev))


def truncated_irfftn_dense(oh: torch.Tensor, out_size: Sequence[int], src_xy: Sequence[int], t_keep: int) -> torch.Tensor:
    """``irfftn(spectrum that is zero outside the kept modes, s=out_size)[..., -t_keep:]`` (unnormalised):
    (b, C, 2mx, 2my, mt) -> (b, C, Xo, Yo, t_keep).  ``src_xy`` = the grid the modes were taken from."""
    Xo, Yo, To = (int(n) for n in out_size)
    Xs, Ys = (int(n) for n in src_xy)
    mx, my, mt = oh.shape[2] // 2, oh.shape[3] // 2, oh.shape[4]
    real = torch.float64 if oh.dtype == torch.complex128 else torch.float32
    dev = oh.device
    g = torch.einsum("bcjkt,jx->bcxkt", oh, _trunc_table("inv_xy", (Xo, Xs, mx), real, dev))
    g = torch.einsum("bcxkt,ky->bcxyt", g, _trunc_table("inv_xy", (Yo, Ys, my), real, dev))
    return (g.real @ _trunc_table("inv_t_re", (To, t_keep, mt), real, dev)
            - g.imag @ _trunc_table("inv_t_im", (To, t_keep, mt), real, dev))
