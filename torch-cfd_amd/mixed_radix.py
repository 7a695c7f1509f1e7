"""Grids the fused kernels do not cover: n = p * 2^k with a small odd factor p (48, 112, 224, 448, ...) on ``CompositeFft``,
every other even n (100, 200, 1000, ...) on ``DenseDft``.

The hand-written transforms are power-of-two Stockham kernels.  The reference accepts any even n
(torch_cfd/equations.py:413-422, ``irfft2`` default size); to keep such grids usable WITHOUT leaving the device or
touching torch.fft, a transform of size n = p * m is decomposed once, by decimation in time / frequency over the odd
factor, into p^2 transforms of size m -- which ARE the HIP kernels -- plus O(p^2 n^2) of element-wise twiddle work
in device tensor ops:

    rfft2 :  y_rs[i, j] = y[p i + r, p j + s]   (p^2 real m x m sub-grids, ONE batched HIP rfft2)
             X[kx, ky]  = sum_{r,s} W^(kx r + ky s) F_rs[kx mod m, ky mod m],        W = exp(-2 pi i / n)
    irfft2:  G_rs[a, b] = W^-(a r + b s) sum_{u,v} X[a + m u, b + m v] exp(+2 pi i (u r + v s) / p)
             y[p i + r, p j + s] = irfft2_m(G_rs)[i, j] / p^2                         (ONE batched HIP irfft2)

``irfft2`` reproduces torch's c2r semantics for spectra that are not Hermitian (the imaginary parts of the DC and
Nyquist COLUMNS are dropped after the transform along x): the two columns are symmetrised first, after which any
exact real inverse gives the same field.

The solver on such a grid runs the stage loop of ``autograd.py`` (device tensor ops around these transforms): the
same arithmetic as the fused kernels, ~100x more launches -- a compatibility path, not a fast one.
"""
from __future__ import annotations

import math
from typing import Optional

import torch


def odd_factor_split(n: int):
    """(p, m) with n = p * m, p odd > 1, m a power of two in [8, 2048]; None if n is not of that form."""
    m = n & -n          # largest power of two dividing n
    p = n // m
    if p > 1 and p <= 15 and 8 <= m <= 2048:
        return p, m
    return None


class CompositeFft:
    """rfft2 / irfft2 of (*, n, n) fields, n = p * m, on a power-of-two plan of size m (anything with ``rfft2``,
    ``irfft2``, ``cdtype``, ``rdtype``: a ``_HipPlan`` on the device, a torch.fft stand-in in the CPU tests)."""

    def __init__(self, n: int, p: int, base_plan):
        self.n, self.p, self.msub = n, p, n // p
        self.m = n // 2 + 1                      # columns of the half spectrum (the name the other plans use)
        self.base = base_plan
        self.cdtype, self.rdtype = base_plan.cdtype, base_plan.rdtype
        self._tables = {}

    # -- tables (per device): W^(k r) for k < n, r < p ; exp(+2 pi i u r / p) ; index maps
    def _t(self, device):
        t = self._tables.get(device)
        if t is None:
            n, p, m = self.n, self.p, self.msub
            k = torch.arange(n, device=device, dtype=torch.float64)
            r = torch.arange(p, device=device, dtype=torch.float64)
            ang = -2 * math.pi * (r[:, None] * k[None, :]) / n
            tw = torch.polar(torch.ones_like(ang), ang).to(self.cdtype)                  # (p, n)  W^(k r)
            angp = 2 * math.pi * (r[:, None] * r[None, :]) / p
            wp = torch.polar(torch.ones_like(angp), angp).to(self.cdtype)                # (p, p)  exp(+2 pi i u r / p)
            t = {"tw": tw, "wp": wp,
                 "kmod": torch.arange(n, device=device) % m,                             # k -> k mod m
                 "neg_m": (-torch.arange(m, device=device)) % m, "neg_n": (-torch.arange(n, device=device)) % n}
            self._tables[device] = t
        return t

    def rfft2(self, y: torch.Tensor) -> torch.Tensor:
        n, p, m = self.n, self.p, self.msub
        lead = y.shape[:-2]
        t = self._t(y.device)
        sub = y.reshape(-1, m, p, m, p).permute(0, 2, 4, 1, 3).contiguous()              # (B, r, s, i, j)
        B = sub.shape[0]
        h = self.base.rfft2(sub.reshape(B * p * p, m, m)).reshape(B, p, p, m, m // 2 + 1)
        # full spectrum along the last axis of every sub-transform: F[a, b] = conj F[-a, m - b]
        upper = h[..., t["neg_m"], :][..., 1: m // 2].flip(-1).conj()
        full = torch.cat([h, upper], dim=-1)                                             # (B, r, s, a, b), b < m
        ky = t["kmod"][: n // 2 + 1]
        # sum over s:  C[B, r, a, ky] = sum_s W^(ky s) F_rs[a, ky mod m]
        c = (full[..., ky] * t["tw"][None, None, :, None, : n // 2 + 1]).sum(dim=2)
        # sum over r:  X[kx, ky] = sum_r W^(kx r) C[r, kx mod m, ky]
        x = (c[:, :, t["kmod"], :] * t["tw"][None, :, :, None]).sum(dim=1)
        return x.reshape(*lead, n, n // 2 + 1)

    def irfft2(self, xh: torch.Tensor) -> torch.Tensor:
        n, p, m = self.n, self.p, self.msub
        lead = xh.shape[:-2]
        t = self._t(xh.device)
        x = xh.reshape(-1, n, n // 2 + 1).to(self.cdtype)
        B = x.shape[0]
        # torch's c2r drops Im of the DC / Nyquist columns AFTER the transform along x: symmetrise those two columns
        x = x.clone()
        for col in (0, n // 2):
            x[:, :, col] = 0.5 * (x[:, :, col] + x[:, t["neg_n"], col].conj())
        # Hermitian completion along ky: X[kx, ky] = conj X[-kx, n - ky] for ky > n/2
        upper = x[:, t["neg_n"], :][..., 1: n // 2].flip(-1).conj()
        full = torch.cat([x, upper], dim=-1).reshape(B, p, m, p, m)                      # (B, u, a, v, b): kx = a + m u
        half = full[..., : m // 2 + 1]                                                   # b <= m/2 is all irfft2_m reads
        # G_rs[a, b] = W^-(a r + b s) sum_{u,v} X[a + m u, b + m v] e^{+2 pi i (u r + v s) / p}
        g = torch.einsum("buavc,ur,vs->brsac", half, t["wp"], t["wp"])
        twa = t["tw"][:, :m].conj()                                                      # (r, a)  W^-(a r)
        twb = t["tw"][:, : m // 2 + 1].conj()                                            # (s, b)
        g = g * twa[None, :, None, :, None] * twb[None, None, :, None, :]
        sub = self.base.irfft2(g.reshape(B * p * p, m, m // 2 + 1).contiguous()).reshape(B, p, p, m, m) / (p * p)
        y = sub.permute(0, 3, 1, 4, 2).reshape(B, n, n)                                  # y[p i + r, p j + s]
        return y.reshape(*lead, n, n)


class DenseDft:
    """rfft2 / irfft2 of (*, n, n) fields for ANY even n as dense transforms: three matrix products with the DFT matrices
    on the device (rocBLAS GEMMs -- the one place a library GEMM is the right tool: O(n^3) per field, but on the fp64
    matrix pipe a 1000^2 field is ~16 GFLOP = a fraction of a millisecond, and no radix schedule exists for, say, n = 2 * 499).

    Serves the even sizes that neither the fused kernels (2^k, 3 * 2^k, 5 * 2^k) nor ``CompositeFft`` (odd factor <= 15)
    cover: 100, 200, 250, 1000, ...  The reference takes any even n (torch_cfd/equations.py:413-422).  Angles are reduced
    with integer arithmetic (j k mod n) before the sine / cosine, so the tables are accurate to the last bit for every n.

        rfft2 :  H = y Cy            (n x n real) (n x m complex)          Cy[j, k] = exp(-2 pi i j k / n), m = n/2 + 1
                 X = Fx H            (n x n complex) (n x m complex)       Fx[a, j] = exp(-2 pi i a j / n)
        irfft2:  G = conj(Fx) X      inverse along x
                 y = Re(G) Ar - Im(G) Ai,   Ar[k, j] = c_k cos(2 pi j k / n) / n^2,  Ai[k, j] = c_k sin(2 pi j k / n) / n^2
    with c = 1 on the DC and Nyquist columns, 2 elsewhere -- torch's c2r semantics: the imaginary parts of those two
    columns are dropped after the transform along x (their sine rows are exactly zero here)."""

    p = 0      # no odd-factor split: TensorOpPlan.info() reports "composite": 0

    def __init__(self, n: int, cdtype: torch.dtype):
        if n < 4 or n % 2 or n > 4096:
            raise ValueError(f"dense transforms cover even 4 <= n <= 4096, got {n}")
        self.n, self.m = n, n // 2 + 1
        self.cdtype = cdtype
        self.rdtype = torch.float64 if cdtype == torch.complex128 else torch.float32
        self._tables = {}

    def _t(self, device):
        t = self._tables.get(device)
        if t is None:
            n, m = self.n, self.m
            j = torch.arange(n, device=device)
            ang_full = (2 * math.pi / n) * ((j[:, None] * j[None, :]) % n).to(torch.float64)           # (n, n)
            ang_half = ang_full[:, :m]                                                                 # (j, k)
            fx = torch.polar(torch.ones_like(ang_full), -ang_full)                                     # exp(-2 pi i a j / n)
            c = torch.full((m,), 2.0, dtype=torch.float64, device=device)
            c[0] = c[-1] = 1.0
            ar = (c[:, None] * torch.cos(ang_half.t())) / float(n * n)                                 # (k, j)
            ai = (c[:, None] * torch.sin(ang_half.t())) / float(n * n)
            ai[0].zero_()
            ai[-1].zero_()
            t = {"cy_re": torch.cos(ang_half).to(self.rdtype), "cy_im": (-torch.sin(ang_half)).to(self.rdtype),
                 "fx": fx.to(self.cdtype), "fxc": fx.conj().resolve_conj().to(self.cdtype),
                 "ar": ar.to(self.rdtype), "ai": ai.to(self.rdtype)}
            self._tables[device] = t
        return t

    def rfft2(self, y: torch.Tensor) -> torch.Tensor:
        n, m = self.n, self.m
        lead = y.shape[:-2]
        t = self._t(y.device)
        y3 = y.reshape(-1, n, n).to(self.rdtype)
        h = torch.complex(y3 @ t["cy_re"], y3 @ t["cy_im"])                                            # (B, n, m)
        return (t["fx"] @ h).reshape(*lead, n, m)

    def irfft2(self, xh: torch.Tensor) -> torch.Tensor:
        n, m = self.n, self.m
        lead = xh.shape[:-2]
        t = self._t(xh.device)
        g = t["fxc"] @ xh.reshape(-1, n, m).to(self.cdtype)                                            # (B, n, m)
        y = g.real @ t["ar"] - g.imag @ t["ai"]
        return y.reshape(*lead, n, n)


class TensorOpPlan:
    """The operations ``NavierStokes2DSpectral`` asks of a plan (step / explicit terms / residual sweep / velocity /
    transforms), carried out with the stage loop of ``autograd.py`` around a ``CompositeFft``."""

    handle = None

    def __init__(self, op, fft: CompositeFft, device, forcing_hat: Optional[torch.Tensor]):
        self.op, self.fft, self.device = op, fft, torch.device(device)
        self.n, self.m, self.cdtype, self.rdtype = fft.n, fft.m, fft.cdtype, fft.rdtype
        self.forcing = None if forcing_hat is None else forcing_hat.to(device=device, dtype=fft.cdtype)

    def info(self):
        return {"separable": 0, "sparse_forcing": 0, "keep_cols": 0, "split": 0, "rows_kernel": 0, "composite": self.fft.p,
                "dense_dft": int(isinstance(self.fft, DenseDft))}

    def _w(self, w):
        if not w.is_cuda and self.device.type == "cuda":
            from . import _lib

            raise _lib.TcfdError("expected a HIP device tensor (torch-cfd_amd has no CPU fallback)")
        return w.detach().to(self.cdtype).reshape(-1, self.n, self.m)

    def rfft2(self, x):
        return self.fft.rfft2(x.detach().to(self.rdtype))

    def irfft2(self, xh):
        return self.fft.irfft2(xh.detach())

    def explicit_terms(self, w):
        from . import autograd as ad

        with torch.no_grad():
            return ad.explicit_terms(self.op, self.fft, self._w(w), self.forcing).reshape(w.shape)

    def step(self, w, beta, gdt, mu, steps, inv_total_dt, want_dwdt=True, fa=None, mu_den=None, base0=None):
        from . import autograd as ad

        sched = {"beta": beta, "gdt": gdt, "mu": mu, "fa": fa, "mu_den": mu_den, "base0": base0}
        w3 = self._w(w)
        with torch.no_grad():
            out = ad.scheduled_steps(self.op, self.fft, w3, steps, sched, self.forcing)
        dwdt = (out - w3) * inv_total_dt if want_dwdt else None
        return out, dwdt

    def velocity(self, w):
        w3 = self._w(w)
        kx, ky = self.op.kx.to(w3.device), self.op.ky.to(w3.device)
        lap = (-4 * torch.pi**2 * (kx**2 + ky**2)).clone()
        lap[..., 0, 0] = 1
        psi = -w3 / lap
        shape = w.shape     # same shapes as the power-of-two plan and the reference: (n, m) in -> (n, m) out
        return ((2j * torch.pi * ky * psi).reshape(shape), (-2j * torch.pi * kx * psi).reshape(shape)), psi.reshape(shape)

    def stream_residual(self, w, wt, want_psi=True, want_res=True):
        psi = res = None
        if want_psi:
            _, psi = self.velocity(w)
        if want_res:
            w3 = self._w(w)
            res = self._w(wt) - self.explicit_terms(w3) - self.op.linear_term.to(w3.device) * w3
        return psi, res
