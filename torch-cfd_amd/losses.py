"""Fourier-domain Sobolev loss of the SFNO training path (reference: fno/losses.py:199-315) on the HIP kernels of
csrc/tcfd_loss.hip: the forward value in three launches on the time-last tensors in place, and -- under autograd -- its gradient with
respect to the prediction in three more (``tcfd_sobolev_loss_backward``).  Shapes outside the fused kernels' cover compose the loss
from the HIP rfft2 (``autograd.Rfft2``) and a one-pass weighted reduction."""
import ctypes
import math
import os
from typing import Dict

import torch
import torch.nn as nn

from . import _lib
from .autograd import Rfft2 as _Rfft2Fn   # rfft2 on the HIP kernels with its hand-written adjoint

_LOSS_PLANS: Dict[tuple, ctypes.c_void_p] = {}   # (n, precision, device) -> tcfd_loss_plan (a twiddle table; lives with the process)


def hip_weighted_sqnorm(zh: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """``(|zh|^2 * w2).sum(dim=(-2, -1))`` for half spectra ``zh`` (*, n, m) and real weights ``w2`` (n, m) in one pass
    over the spectrum (``tcfd_weighted_sqnorm``; double accumulation, the result comes back in ``w2.dtype``)."""
    if not zh.is_cuda:
        raise _lib.TcfdError("expected a HIP device tensor (torch-cfd_amd has no CPU fallback)")
    lead, elems = zh.shape[:-2], zh.shape[-2] * zh.shape[-1]
    if tuple(w2.shape) != tuple(zh.shape[-2:]):
        raise ValueError(f"weights {tuple(w2.shape)} for spectra {tuple(zh.shape[-2:])}")
    cdt = zh.dtype
    rdt = torch.float64 if cdt == torch.complex128 else torch.float32
    zc = zh.contiguous()
    wc = w2.to(device=zh.device, dtype=rdt).contiguous()
    batch = max(int(zc.numel() // elems), 1)
    blocks = max(1, min(64, (elems + 4095) // 4096))
    partial = torch.empty(batch, blocks, dtype=torch.float64, device=zh.device)
    with torch.cuda.device(zh.device):
        _lib.check(_lib.load().tcfd_weighted_sqnorm(
            zc.data_ptr(), wc.data_ptr(), partial.data_ptr(), batch, elems, blocks,
            _lib.TCFD_C128 if cdt == torch.complex128 else _lib.TCFD_C64,
            ctypes.c_void_p(torch.cuda.current_stream(zh.device).cuda_stream)), "tcfd_weighted_sqnorm")
    return partial.sum(dim=-1).to(w2.dtype).reshape(lead)


def _fused_forward(plan, xc, yc, w2, ws, flags, sums):
    """``tcfd_sobolev_loss`` on contiguous time-last tensors; ``sums``: (nfields, batch, nt) doubles kept for the backward pass."""
    nf, relative, mesh, tavg, red = flags
    bsz, nt = xc.shape[0], xc.shape[-1]
    out = torch.empty((), dtype=xc.dtype, device=xc.device)
    with torch.cuda.device(xc.device):
        _lib.check(_lib.load().tcfd_sobolev_loss(plan, xc.data_ptr(), yc.data_ptr() if yc is not None else None, w2.data_ptr(), bsz, nt,
                                                 nf, relative, mesh, tavg, red, out.data_ptr(),
                                                 sums.data_ptr() if sums is not None else None, ws.data_ptr(), ws.numel(),
                                                 ctypes.c_void_p(torch.cuda.current_stream(xc.device).cuda_stream)), "tcfd_sobolev_loss")
    return out


class _FusedLossFn(torch.autograd.Function):
    """The fused loss as one autograd node: forward = the three launches of ``tcfd_sobolev_loss`` (per-time sums kept), backward =
    the three of ``tcfd_sobolev_loss_backward`` -- d = x - y is transformed again rather than 94 MB of half spectra kept across
    the training step.  Replaces (round 4) two permuted copies, an rfft2 with its adjoint and eight elementwise kernels over the
    spectrum: 0.9 -> 0.3 ms of a config-5 training step."""

    @staticmethod
    def forward(ctx, x, yc, plan, w2, wf, ws, flags, module):
        xc = x.detach()           # contiguous: SobolevLoss._fused hands over the contiguous form (a view of the caller's tensor when it is)
        sums = torch.empty(flags[0] * xc.shape[0] * xc.shape[-1], dtype=torch.float64, device=xc.device)
        out = _fused_forward(plan, xc, yc, w2, ws, flags, sums)
        # x itself, not its detached twin: a backward pass under create_graph=True differentiates through it again
        ctx.save_for_backward(x, yc if yc is not None else xc.new_empty(0), wf, sums)
        ctx.cfg = (plan, ws, flags, yc is not None, module)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, yc, wf, sums = ctx.saved_tensors
        plan, ws, flags, has_y, module = ctx.cfg
        if torch.is_grad_enabled():
            # create_graph=True (gradient penalties, Hessian-vector products through the loss): the raw-pointer launches below
            # would return a constant.  Form the gradient from the composed loss -- Rfft2 (autograd.py, differentiable any number
            # of times) plus tensor operations -- as FusedExplicitTerms.backward does for the solver, and as the reference's
            # pure-torch loss allows (fno/losses.py:263-315).
            with torch.enable_grad():
                xin = x if x.requires_grad else x.detach().requires_grad_(True)
                loss = module._composed(xin, yc if has_y else None)
                (grad,) = torch.autograd.grad(loss, xin, gout.to(loss.dtype), create_graph=True)
            return grad, None, None, None, None, None, None, None
        xc = x.detach()
        nf, relative, mesh, tavg, red = flags
        g = gout.detach().to(xc.dtype).contiguous()
        grad = torch.empty_like(xc)
        with torch.cuda.device(xc.device):
            _lib.check(_lib.load().tcfd_sobolev_loss_backward(
                plan, xc.data_ptr(), yc.data_ptr() if has_y else None, wf.data_ptr(), sums.data_ptr(), g.data_ptr(), xc.shape[0],
                xc.shape[-1], nf, relative, mesh, tavg, red, grad.data_ptr(), ws.data_ptr(), ws.numel(),
                ctypes.c_void_p(torch.cuda.current_stream(xc.device).cuda_stream)), "tcfd_sobolev_loss_backward")
        return grad, None, None, None, None, None, None, None


# workspaces of the fused loss, per (device, bytes needed rounded up): outside the modules (a device tensor in a plain module
# attribute rides along with copy.deepcopy / torch.save(module), is not moved by .to() and would pin a CUDA graph's private pool
# if first allocated during capture)
_LOSS_WORKSPACES: Dict[torch.device, torch.Tensor] = {}


def _loss_workspace(device: torch.device, need: int) -> torch.Tensor:
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(need, dtype=torch.uint8, device=device)     # belongs to the capture, never cached
    ws = _LOSS_WORKSPACES.get(device)
    if ws is None or ws.numel() < need:
        _LOSS_WORKSPACES.pop(device, None)
        ws = _LOSS_WORKSPACES[device] = torch.empty(need, dtype=torch.uint8, device=device)
    return ws


class SobolevLoss(nn.Module):
    """Fourier-domain weighted norm of (x - y), fno/losses.py:199-315, including ``freq_cutoff`` (wavenumbers above it
    are replaced by inf for negative orders and by 0 otherwise, exactly as the reference's mesh does) and every
    ``fft_norm``.

    Copies the code's behaviour, not its comment: for ``norm_order == 0`` the multiplier is
    sqrt(alpha + 4 pi^2 |k|^2) itself, not 1.  The 2-D transforms over dims (1, 2) of the
    time-last tensors run on the HIP rfft2 kernels (one Hermitian-weighted half spectrum per time slice
    instead of the reference's full complex fftn); everything else is a handful of device reductions."""

    def __init__(self, n_grid: int = 256, time_average: bool = True, reduction: bool = True, mesh_weighted: bool = True,
                 relative: bool = False, inp_time_last: bool = True, freq_cutoff: int = None, norm_order: float = -1,
                 alpha: float = 0.1, fft_norm: str = "backward", diam: float = 1, debug: bool = False):
        super().__init__()
        if fft_norm not in (None, "backward", "ortho", "forward"):
            raise ValueError(f"unknown fft norm {fft_norm!r}")
        # |fftn(z, norm)|^2 = |fftn(z)|^2 / n^2 ("ortho") or / n^4 ("forward"): a scalar on the squared norms
        self.fft_norm = fft_norm
        self._sq_scale = {None: 1.0, "backward": 1.0, "ortho": 1.0 / n_grid**2, "forward": 1.0 / n_grid**4}[fft_norm]
        self.relative, self.time_average, self.reduction = relative, time_average, reduction
        self.mesh_weighted, self.norm_order, self.alpha = mesh_weighted, norm_order, alpha
        self.inp_time_last, self.n_grid, self.diam = inp_time_last, n_grid, diam
        n = n_grid
        k = torch.fft.fftfreq(n, d=diam / n)
        kx, ky = torch.meshgrid([k, k], indexing="ij")
        cutoff = (n // 2 + 1 if freq_cutoff is None else freq_cutoff) / diam
        fill = math.inf if norm_order < 0 else 0.0
        kx = kx.clone().masked_fill(kx.abs() > cutoff, fill)
        ky = ky.clone().masked_fill(ky.abs() > cutoff, fill)
        weight = alpha + 4 * (torch.pi) ** 2 * (kx**2 + ky**2)
        self.register_buffer("kx", kx[None, :, :, None])
        self.register_buffer("ky", ky[None, :, :, None])
        self.register_buffer("weight", weight[None, :, :, None])

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_w2_cache", None)      # a device-side table rebuilt on demand: not part of a pickled / deep-copied module
        return state

    def _half_spectrum_weights(self, device, dtype):
        """(n, n/2+1) table: multiplier^2 x Hermitian multiplicity x fft-norm scale.  Built once per (device, dtype) and
        state of the ``weight`` buffer -- it took ~20 small launches per call, more than the loss kernels themselves."""
        key = (torch.device(device), dtype, self.weight.data_ptr(), self.weight._version, self.norm_order, self._sq_scale)
        cached = self.__dict__.get("_w2_cache")
        if cached is not None and cached[0] == key:
            return cached[1]
        n = self.n_grid
        # sqrt and power in the BUFFER's precision, as the reference forms its multiplier (losses.py:279-289) before the
        # product with the spectrum promotes it: a float32 module on float64 data uses float32-rounded weights there too
        # (and on the CPU, once: a device pow differs from the CPU's in the last float32 bit, 5e-10 on an order -1 loss)
        w = torch.sqrt(self.weight[0, :, : n // 2 + 1, 0].detach().cpu())
        w = (w ** (self.norm_order / 2) if self.norm_order != 0 else w).to(device=device, dtype=dtype)
        herm = torch.full((n // 2 + 1,), 2.0, dtype=dtype, device=device)  # |X[k]|^2 counted twice except DC/Nyquist
        herm[0] = 1.0
        herm[-1] = 1.0
        w2 = (w**2 * herm * self._sq_scale).contiguous()
        self._w2_cache = (key, w2, (w**2 * self._sq_scale).contiguous())    # ... and without the multiplicity (backward pass)
        return w2

    def _fused(self, x, y):
        """The loss in three launches on the time-last tensors in place (``tcfd_sobolev_loss``, csrc/tcfd_loss.hip), or None
        when this call is outside its cover (a target that needs a gradient, a grid off the FFT kernels -- 2^k in [16, 1024],
        3 * 2^k in [96, 768], 5 * 2^k in [80, 640] --, more time steps than one workgroup transforms).  With a prediction that
        needs a gradient the same kernels run inside ``_FusedLossFn``, whose backward is ``tcfd_sobolev_loss_backward``."""
        if os.environ.get("TCFD_LOSS_FUSED", "1") == "0" or not x.is_cuda or x.dtype not in (torch.float32, torch.float64):
            return None
        wants_grad = torch.is_grad_enabled() and x.requires_grad
        if torch.is_grad_enabled() and y is not None and y.requires_grad:
            return None                                   # a target that needs a gradient: the composed path below
        if wants_grad and os.environ.get("TCFD_LOSS_FUSED_BWD", "1") == "0":
            return None
        bsz, n, n2, nt = x.shape
        if (n != n2 or not ((16 <= n <= 1024 and (n & (n - 1)) == 0) or n in (96, 192, 384, 768, 80, 160, 320, 640)) or bsz == 0
                or (y is not None and (y.shape != x.shape or y.dtype != x.dtype))):
            return None
        lib = _lib.load()
        code = _lib.TCFD_C128 if x.dtype == torch.float64 else _lib.TCFD_C64
        pkey = (n, code, x.device)
        plan = _LOSS_PLANS.get(pkey)
        if plan is None:
            handle = ctypes.c_void_p()
            with torch.cuda.device(x.device):
                _lib.check(lib.tcfd_loss_plan_create(ctypes.byref(handle), n, code), "tcfd_loss_plan_create")
            plan = _LOSS_PLANS[pkey] = handle
        nf = 2 if (self.relative and y is not None) else 1
        if not lib.tcfd_sobolev_loss_supported(plan, nt, nf):
            return None
        xc = x.contiguous()
        yc = y.contiguous() if y is not None else None
        w2 = self._half_spectrum_weights(x.device, x.dtype)
        ws = _loss_workspace(x.device, lib.tcfd_loss_workspace_bytes(plan, bsz, nt, nf))
        flags = (nf, int(bool(self.relative and y is not None)),
                 (2 if torch.get_default_dtype() == torch.float32 else 1) if self.mesh_weighted else 0,
                 int(bool(self.time_average)), int(bool(self.reduction)))
        if wants_grad:
            return _FusedLossFn.apply(xc, yc, plan, w2, self._w2_cache[2], ws, flags, self)
        return _fused_forward(plan, xc, yc, w2, ws, flags, None)

    def forward(self, x, y=None):
        if not self.inp_time_last:
            x = x.permute(0, 2, 3, 1)
            y = y.permute(0, 2, 3, 1) if y is not None else None
        bsz, n, _, nt = x.shape
        if n != self.n_grid:
            raise ValueError(f"grid {n} != n_grid {self.n_grid}")
        # relative loss without a target: the reference divides by the norm of its all-zero y (losses.py:283-299) -- inf (nan
        # for x = 0), reproduced by dividing the plain norm by zero
        no_target = 0.0 if (self.relative and y is None) else None
        fused = self._fused(x, y)
        if fused is not None:
            return fused if no_target is None else fused / no_target
        loss = self._composed(x, y)
        return loss if no_target is None else loss / no_target

    def _composed(self, x, y=None):
        """The loss of time-last x (and y) composed from the HIP rfft2 (``autograd.Rfft2`` under autograd: differentiable any
        number of times) and tensor operations: what runs outside the fused kernels' cover, and what the fused node's backward
        differentiates when a graph of the gradient itself is asked for."""
        from .equations import fft_plan

        bsz, n, _, nt = x.shape
        plan = fft_plan(n, torch.complex64 if x.dtype == torch.float32 else torch.complex128, x.device, self.diam)
        w2 = self._half_spectrum_weights(x.device, x.dtype)

        def sq_norms(z):  # (b, n, n, t) -> (b, t): || w * fft2(z_t) ||_F^2 via the half spectrum
            zt = z.permute(0, 3, 1, 2).contiguous()
            if torch.is_grad_enabled() and zt.requires_grad:
                zh = _Rfft2Fn.apply(zt, plan)
                return ((zh.real**2 + zh.imag**2) * w2).sum(dim=(-2, -1))
            zh = plan.rfft2(zt)
            return hip_weighted_sqnorm(zh, w2)

        diff = sq_norms(x if y is None else x - y)  # the transform is linear: one rfft2 of the difference
        loss = diff.sum(dim=-1).sqrt()
        if self.relative and y is not None:
            yn = sq_norms(y).sum(dim=-1).sqrt()
        else:
            # the reference's unit norms are torch.ones(bsz) in the DEFAULT dtype (losses.py:297): under a float32 default, 1 / n
            # is rounded to float32 before it divides a float64 loss (1.5e-8 at n = 80; exact when n is a power of two)
            yn = torch.ones(bsz, device=x.device, dtype=torch.get_default_dtype())
        yn = (yn / n if self.mesh_weighted else yn).to(x.dtype)
        loss = loss / yn
        loss = loss / math.sqrt(nt) if self.time_average else loss
        loss = loss.mean(0) if self.reduction else loss.sum(0)
        return loss / n if self.mesh_weighted else loss
