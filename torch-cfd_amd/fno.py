"""FNO / SFNO spectral layers on the HIP pruned-transform + MFMA-contraction kernels.

Drop-in operator API for (same class names, constructor arguments, parameter
names / shapes and ``state_dict`` keys):

  * ``SpectralConv``      fno/base.py:114-237   (template: rfftn -> spectral_conv -> irfftn)
  * ``SpectralConvS``     fno/sfno.py:331-394
  * ``SpectralConvT``     fno/sfno.py:397-457   (time padding / arbitrary output steps)
  * ``SpectralConv3d``    fno/fno3d.py:19-116   (4 complex ``weights1..4`` parameters)
  * ``LayerNormnd``, ``PointwiseFFN``, ``SpaceTimePositionalEncoding``, ``HelmholtzProjection``,
    ``LiftingOperator``, ``OutConv``, ``SFNO``   fno/base.py:61-111, fno/sfno.py:25-328, 460-620

The spectral convolutions call ``tcfd_fno_spectral_conv`` (include/tcfd.h): five
kernels that read and write the (b, C, X, Y, T) activations exactly once instead
of the reference's full rfftn / zero-filled spectrum / irfftn.  fp32, HIP device
tensors, X and Y powers of two; anything else raises (no fallback).  Under autograd the
spectral convolutions run a hand-written backward on the same kernels (``hip_spectral_conv_autograd``);
the pointwise blocks keep their HIP forward and recompute the block with torch einsums in the
backward (``_PointwiseFn``).
The pointwise layers around them (1x1x1 convolutions, GroupNorm, activations) are
ordinary torch modules running on the same device.
"""
from __future__ import annotations

import ctypes
import functools
import os
import math
import weakref
from typing import Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

conv_dict = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}
ActivationType = Union[str]


# ----------------------------------------------------------------------------- HIP plan cache
class _FnoPlan:
    def __init__(self, key, device, real: torch.dtype = torch.float32):
        """key = (X, Y, T_in, t_pad, T_out, mx, my, mt[, Xs, Ys]); the two extra entries make it the inverse plan of a
        RESAMPLING layer (spectrum taken on an Xs x Ys grid, transformed onto X x Y).  ``real`` is the precision of the
        plan's tables and of every array passed with it: float32 (complex64 spectra) or float64 (complex128)."""
        X, Y, T_in, t_pad, T_out, mx, my, mt = key[:8]
        self.lib = _lib.load()
        self.key = tuple(key[:8])
        self.device = torch.device(device)
        self.real = real
        self.cplx = torch.complex128 if real == torch.float64 else torch.complex64
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            Xs, Ys = (key[8], key[9]) if len(key) == 10 else (X, Y)
            rc = self.lib.tcfd_fno_plan_create_dtype(ctypes.byref(handle), X, Y, T_in, t_pad, T_out, mx, my, mt, Xs, Ys,
                                                     _lib.TCFD_C128 if real == torch.float64 else _lib.TCFD_C64)
        _lib.check(rc, "tcfd_fno_plan_create_dtype")
        self.handle = handle
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._fin = weakref.finalize(self, self.lib.tcfd_fno_plan_destroy, handle)

    def workspace(self, b, ci, co):
        k = (b, ci, co)
        ws = self._ws.get(k)
        if ws is None:
            n = self.lib.tcfd_fno_workspace_bytes(self.handle, b, ci, co)
            self._ws = {k: torch.empty(n, dtype=torch.uint8, device=self.device)}
            ws = self._ws[k]
        return ws


_PLANS: Dict[tuple, _FnoPlan] = {}


_WARNED = set()


def _note_torch_modules(what: str):
    """The fused pointwise kernels cover every width <= 32 and 36 / 40 / 48 / 64 (any expansion); other channel counts run
    the layer's own torch modules ON THE DEVICE (never the oracle / CPU) -- said once, not silently."""
    if what not in _WARNED:
        _WARNED.add(what)
        import warnings

        warnings.warn(f"torch-cfd_amd: {what} is not instantiated in the fused HIP pointwise kernels; "
                      "this layer uses its torch modules on the device (slower, same result)", stacklevel=3)


def _real_of(dtype: torch.dtype) -> torch.dtype:
    return torch.float64 if dtype in (torch.float64, torch.complex128) else torch.float32


def _plan(key, device, real: torch.dtype = torch.float32) -> _FnoPlan:
    full = key + (torch.device(device), real)
    p = _PLANS.get(full)
    if p is None:
        p = _FnoPlan(key, device, real)
        _PLANS[full] = p
    return p


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * 4)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _norm_scales(norm: str, n_in: int, n_out: int) -> Tuple[float, float]:
    if norm in (None, "backward"):
        return 1.0, 1.0 / n_out
    if norm == "ortho":
        return 1.0 / math.sqrt(n_in), 1.0 / math.sqrt(n_out)
    if norm == "forward":
        return 1.0 / n_in, 1.0
    raise ValueError(f"unknown fft norm {norm!r}")


def _pow2_xy(X: int, Y: int) -> bool:
    """Spatial sizes of the FFT kernels (k_fwd_ty2 / k_x / k_inv_ty2): powers of two in [8, 1024]."""
    return all(8 <= n <= 1024 and (n & (n - 1)) == 0 for n in (X, Y))


def _fused_xy(X: int, Y: int) -> bool:
    """Spatial sizes the library's transform kernels cover (tcfd_fno_plan_create): powers of two on the FFT kernels, every
    other size in [4, 1024] on the pruned direct-DFT kernels (k_fwd_ty_dft / k_x_dft / k_inv_ty_dft).  ``TCFD_FNO_DENSE=1``
    sends the non-power-of-two sizes through the dense GEMM transforms of dense_fft.py instead (cross-check)."""
    if os.environ.get("TCFD_FNO_DENSE", "0") == "1":
        return _pow2_xy(X, Y)
    return all(4 <= n <= 1024 for n in (X, Y))


def _fft_len(n: int) -> bool:
    """Lengths of the FFT kernels (tcfd_fno.hip fft_len): 2^k in [8, 1024], 3 * 2^k in [96, 768], 5 * 2^k in [80, 640]."""
    return (8 <= n <= 1024 and (n & (n - 1)) == 0) or n in (96, 192, 384, 768, 80, 160, 320, 640)


def _library_takes(X: int, Y: int, T: int, t_pad: int, t_out: int, modes, t_keep: int, device, real=torch.float32) -> bool:
    """Whether the library's transform kernels take this call (``tcfd_fno_plan_supports``): FFT lengths always; the pruned
    direct-DFT kernels of the other sizes hold a (Y x time) slab in LDS and know at most 16 time modes -- beyond that the
    layers run the dense GEMM transforms (``dense_spectral_conv``), as they do for sizes outside [4, 1024]."""
    if not _fused_xy(X, Y):
        return False
    if _fft_len(Y) and os.environ.get("TCFD_FNO_DFT", "0") != "1":
        return True
    plan = _plan((X, Y, T, t_pad, t_out) + tuple(modes), device, real)
    return bool(plan.lib.tcfd_fno_plan_supports(plan.handle, int(t_keep)))


def dense_spectral_conv(v: torch.Tensor, weights, bias, delta: float, modes, t_pad: int = 0, t_out: Optional[int] = None,
                        t_keep: Optional[int] = None, norm: str = "backward", out_xy=None, post=None,
                        use_mfma: bool = True) -> torch.Tensor:
    """The spectral convolution for ANY spatial size (the reference's rfftn / irfftn take any mesh, fno/base.py:229-237): the
    pruned transforms as thin matrix products with the kept rows of the DFT matrices (``dense_fft.truncated_*``: three GEMMs
    per direction, rocBLAS), the 4-corner contraction on the same MFMA kernel as the fused path.  Differentiable (matmuls +
    ``_ContractFn``), so it is also the gradient path of a spatially resampled layer.  ``out_xy``: output grid when the layer
    resamples in space (the high-frequency block keeps its array indices, as torch's ``irfftn(s=...)`` does)."""
    from .dense_fft import truncated_irfftn_dense, truncated_rfftn_dense

    if not v.is_cuda:
        raise _lib.TcfdError("expected a HIP device tensor (torch-cfd_amd has no CPU fallback)")
    if v.dtype not in (torch.float32, torch.float64) or v.dim() != 5:
        raise TypeError(f"expected a real fp32 / fp64 (b, C, X, Y, T) tensor, got {v.dtype} {tuple(v.shape)}")
    b, ci, X, Y, T = v.shape
    Xo, Yo = (X, Y) if out_xy is None else (int(out_xy[0]), int(out_xy[1]))
    t_out = T + t_pad if t_out is None else t_out
    t_keep = t_out if t_keep is None else t_keep
    fs, is_ = _norm_scales(norm, X * Y * (T + t_pad), Xo * Yo * t_out)
    params = list(weights) + (list(bias) if bias is not None else [])
    if b == 0:
        return torch.empty(0, weights[0].shape[1], Xo, Yo, t_keep, dtype=v.dtype, device=v.device)
    vh = truncated_rfftn_dense(v, modes, t_pad)
    vh = (vh * fs if fs != 1.0 else vh).contiguous()
    if torch.is_grad_enabled() and (vh.requires_grad or any(p.requires_grad for p in params)):
        oh = _ContractFn.apply(vh, float(delta), tuple(modes), use_mfma, bias is not None, *params)
    else:
        oh = hip_contract(vh, weights, bias, delta, modes, use_mfma=use_mfma)
    if post is not None:
        oh = post(oh).to(vh.dtype)
    out = truncated_irfftn_dense(oh, (Xo, Yo, t_out), (X, Y), t_keep)
    return out * is_ if is_ != 1.0 else out


def hip_spectral_conv(v: torch.Tensor, weights, bias, delta: float, modes, t_pad: int = 0,
                      t_out: Optional[int] = None, t_keep: Optional[int] = None, norm: str = "backward",
                      use_mfma: bool = True) -> torch.Tensor:
    """irfftn(contract(rfftn(left_pad_t(v, t_pad))), s=(X, Y, t_out))[..., -t_keep:] on the HIP kernels.

    v (b, Ci, X, Y, T) fp32 or fp64 HIP tensor (fp64: FNOBase.double(), fno/base.py:342-349 -- the same kernels,
    instantiated for double); weights: 4 tensors (Ci, Co, mx, my, mt) complex or (Ci, Co, mx, my, mt, 2) real of the
    SAME precision; bias: None or 4 tensors (mx, my, mt[, 2])."""
    if not v.is_cuda:
        raise _lib.TcfdError("expected a HIP device tensor (torch-cfd_amd has no CPU fallback)")
    if v.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"the HIP spectral convolution is fp32 or fp64, got {v.dtype}")
    real = v.dtype
    if real == torch.float64 and any(_real_of(w.dtype) != real for w in weights):   # loud, like torch's einsum on mixed dtypes
        raise TypeError("float64 input to a spectral convolution with float32 parameters: call .double() on the layer")
    # (float32 data through a layer whose parameters are float64 -- built under a float64 default dtype -- computes in
    #  float32: the parameters are cast, as in round 2)
    if v.dim() != 5:
        raise ValueError(f"expected (b, C, X, Y, T), got {tuple(v.shape)}")
    b, ci, X, Y, T = v.shape
    mx, my, mt = modes
    co = weights[0].shape[1]
    t_out = T + t_pad if t_out is None else t_out
    t_keep = t_out if t_keep is None else t_keep
    if not _library_takes(X, Y, T, t_pad, t_out, modes, t_keep, v.device, real):
        # sizes outside [4, 1024], or a slab the direct-DFT kernels cannot hold: the same pruned transforms as thin GEMMs
        return dense_spectral_conv(v, weights, bias, delta, modes, t_pad, t_out, t_keep, norm, use_mfma=use_mfma)
    params = list(weights) + (list(bias) if bias is not None else [])
    if torch.is_grad_enabled() and (v.requires_grad or any(p.requires_grad for p in params)):
        return hip_spectral_conv_autograd(v, weights, bias, delta, modes, t_pad, t_out, t_keep, norm, use_mfma)
    v = v.detach().contiguous()
    if b == 0:   # empty batch, like the reference's tensor ops
        return torch.empty(0, co, X, Y, t_keep, dtype=real, device=v.device)

    def as_real(w, shape):
        w = w.detach()
        if w.is_complex():
            w = torch.view_as_real(w)
        w = w.to(real).contiguous()
        if tuple(w.shape) != shape:
            raise ValueError(f"weight/bias shape {tuple(w.shape)} != {shape}")
        return w

    ws_ = [as_real(w, (ci, co, mx, my, mt, 2)) for w in weights]
    bs_ = [as_real(x, (mx, my, mt, 2)) for x in bias] if bias is not None else None
    plan = _plan((X, Y, T, t_pad, t_out, mx, my, mt), v.device, real)
    out = torch.empty(b, co, X, Y, t_keep, dtype=real, device=v.device)
    ws = plan.workspace(b, ci, co)
    fs, is_ = _norm_scales(norm, X * Y * (T + t_pad), X * Y * t_out)
    with torch.cuda.device(v.device):
        rc = plan.lib.tcfd_fno_spectral_conv(
            plan.handle, v.data_ptr(), _ptr_array(ws_), _ptr_array(bs_) if bs_ is not None else None,
            float(delta), out.data_ptr(), b, ci, co, t_keep, fs, is_, 1 if use_mfma else 0,
            ws.data_ptr(), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream(v.device).cuda_stream))
    _lib.check(rc, "tcfd_fno_spectral_conv")
    return out


# ----------------------------------------------------------------------------- the pruned transforms and the contraction, one call each
def hip_truncated_rfftn(v: torch.Tensor, modes, t_pad: int = 0, t_out: Optional[int] = None, norm="backward",
                        scale: Optional[float] = None, kt_scale: Optional[torch.Tensor] = None):
    """Kept modes of rfftn(left_pad_t(v)): (b, C, X, Y, T) real -> (b, C, 2mx, 2my, mt) complex of the same precision
    (fp32 / fp64), plus the plan.  ``scale`` overrides the normalisation factor of ``norm``; ``kt_scale``: (mt,) real device
    tensor, one factor per kept time mode, folded into the transform's t-DFT table (``tcfd_fno_forward_trunc_kt``)."""
    b, c, X, Y, T = v.shape
    mx, my, mt = modes
    t_out = T + t_pad if t_out is None else t_out
    plan = _plan((X, Y, T, t_pad, t_out, mx, my, mt), v.device, _real_of(v.dtype))
    v = v.detach().to(plan.real).contiguous()
    vh = torch.empty(b, c, 2 * mx, 2 * my, mt, dtype=plan.cplx, device=v.device)
    ws = plan.workspace(b, c, c)
    fs, _ = _norm_scales(norm, X * Y * (T + t_pad), X * Y * t_out)
    fs = fs if scale is None else float(scale)
    kts = _kt_factors(kt_scale, mt, plan) if kt_scale is not None else None
    with torch.cuda.device(v.device):
        rc = plan.lib.tcfd_fno_forward_trunc_kt(plan.handle, v.data_ptr(), vh.data_ptr(), b, c, fs, kts.data_ptr() if kts is not None else None,
                                                ws.data_ptr(), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream(v.device).cuda_stream))
    _lib.check(rc, "tcfd_fno_forward_trunc")
    return vh, plan


def _kt_factors(kt_scale: torch.Tensor, mt: int, plan) -> torch.Tensor:
    if kt_scale.numel() != mt or kt_scale.dtype != plan.real or not kt_scale.is_cuda or not kt_scale.is_contiguous():
        raise ValueError(f"kt_scale must be a contiguous ({mt},) device tensor of the transform's precision")
    return kt_scale


def hip_truncated_irfftn(vh: torch.Tensor, plan: "_FnoPlan", t_keep: int, norm="backward",
                         scale: Optional[float] = None, accumulate: Optional[torch.Tensor] = None,
                         add_last: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                         kt_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """irfftn of a spectrum that is zero outside the kept modes: (b, C, 2mx, 2my, mt) -> (b, C, X, Y, t_keep).
    ``accumulate``: a contiguous tensor of the output's shape that the result is ADDED to, in place, by the transform's own
    store loop (``tcfd_fno_inverse_trunc_acc``) -- it is returned.  ``add_last``: (b, C, X, Y[, 1]) added to the LAST kept step
    only, by the same store loop (``tcfd_fno_inverse_trunc_last``).  ``kt_scale``: as in ``hip_truncated_rfftn``."""
    X, Y, T, t_pad, t_out, mx, my, mt = plan.key
    b, c = vh.shape[:2]
    if tuple(vh.shape[2:]) != (2 * mx, 2 * my, mt) or not vh.is_complex():
        raise ValueError(f"expected a complex (b, C, {2 * mx}, {2 * my}, {mt}) truncated spectrum, got {tuple(vh.shape)} {vh.dtype}")
    vh = vh.detach().to(plan.cplx).contiguous()   # e.g. a float64 post-processing table promotes an fp32 layer's spectrum
    if accumulate is not None:
        if (tuple(accumulate.shape) != (b, c, X, Y, t_keep) or accumulate.dtype != plan.real or accumulate.device != vh.device
                or not accumulate.is_contiguous()):
            raise ValueError("accumulate must be a contiguous tensor of the output's shape, precision and device")
        out = accumulate
    elif out is not None:       # a caller-owned buffer (the chunked inference layers reuse one: it then stays in the Infinity Cache)
        if tuple(out.shape) != (b, c, X, Y, t_keep) or out.dtype != plan.real or out.device != vh.device or not out.is_contiguous():
            raise ValueError("out must be a contiguous tensor of the output's shape, precision and device")
    else:
        out = torch.empty(b, c, X, Y, t_keep, dtype=plan.real, device=vh.device)
    ws = plan.workspace(b, c, c)
    _, is_ = _norm_scales(norm, X * Y * (T + t_pad), X * Y * t_out)
    is_ = is_ if scale is None else float(scale)
    if add_last is not None:
        if accumulate is not None or add_last.numel() != b * c * X * Y or add_last.dtype != plan.real or add_last.device != vh.device:
            raise ValueError("add_last must be a (b, C, X, Y) tensor of the output's precision and device, without accumulate")
        last = add_last.detach().contiguous()
    else:
        last = None
    kts = _kt_factors(kt_scale, mt, plan) if kt_scale is not None else None
    with torch.cuda.device(vh.device):
        rc = plan.lib.tcfd_fno_inverse_trunc_kt(plan.handle, vh.data_ptr(), out.data_ptr(),
                                                out.data_ptr() if accumulate is not None else None,
                                                last.data_ptr() if last is not None else None, b, c, t_keep, is_,
                                                kts.data_ptr() if kts is not None else None, ws.data_ptr(), ws.numel(),
                                                ctypes.c_void_p(torch.cuda.current_stream(vh.device).cuda_stream))
    _lib.check(rc, "tcfd_fno_inverse_trunc")
    return out


def hip_contract(vh: torch.Tensor, weights, bias, delta, modes, use_mfma=True) -> torch.Tensor:
    """The 4-corner contraction alone on truncated spectra (b, Ci, 2mx, 2my, mt) complex64 / complex128; the precision of
    the SPECTRUM decides, the weights are cast to it."""
    b, ci = vh.shape[:2]
    mx, my, mt = modes
    co = weights[0].shape[1]
    real = _real_of(vh.dtype)
    cplx = torch.complex128 if real == torch.float64 else torch.complex64
    vh = vh.detach().to(cplx).contiguous()
    f32 = lambda t: (torch.view_as_real(t) if t.is_complex() else t).detach().to(device=vh.device, dtype=real).contiguous()
    ws_ = [f32(w) for w in weights]
    bs_ = [f32(x) for x in bias] if bias is not None else None
    out = torch.empty(b, co, 2 * mx, 2 * my, mt, dtype=cplx, device=vh.device)
    lib = _lib.load()
    with torch.cuda.device(vh.device):
        rc = lib.tcfd_fno_contract(vh.data_ptr(), _ptr_array(ws_), _ptr_array(bs_) if bs_ is not None else None,
                                   float(delta), out.data_ptr(), b, ci, co, mx, my, mt, 1 if use_mfma else 0,
                                   _lib.TCFD_C128 if real == torch.float64 else _lib.TCFD_C64,
                                   ctypes.c_void_p(torch.cuda.current_stream(vh.device).cuda_stream))
    _lib.check(rc, "tcfd_fno_contract")
    return out


_C2R_WEIGHTS: Dict[tuple, torch.Tensor] = {}


def _c2r_weights(mt: int, T: int, device, dtype: torch.dtype = torch.float32, inverse: bool = False) -> torch.Tensor:
    """Multiplicity of the kept time modes in a length-T c2r transform: 1 for kt = 0 and the Nyquist mode, else 2 (``inverse``:
    their reciprocals).  Cached: built in place (``c[0] = 1.0``) it cost a host-to-device copy of a scalar per call -- ten blocking
    copies per training iteration, and a capture-breaking one under a graph.  The backward transforms take the table as their
    per-time-mode factor (``kt_scale``): no pass over the spectrum."""
    key = (mt, T, torch.device(device), _real_of(dtype), inverse)
    c = _C2R_WEIGHTS.get(key)
    if c is None:
        host = [0.5 if inverse else 2.0] * mt
        host[0] = 1.0
        if T % 2 == 0 and T // 2 < mt:
            host[T // 2] = 1.0
        c = _C2R_WEIGHTS[key] = torch.tensor(host, dtype=_real_of(dtype)).to(device)
    return c


def _fwd_trunc_vjp(z, cfg, accumulate=None, add_last=None):
    """F^T(z) for the truncated forward transform F (cfg of ``_FwdTruncFn``); ``accumulate``: added to, in place; ``add_last``:
    (b, c, X, Y[, 1]) added to the last time step only."""
    (b, c, X, Y, T), modes, t_pad, t_out, norm = cfg
    Tp = T + t_pad
    fs, _ = _norm_scales(norm, X * Y * Tp, X * Y * t_out)
    plan = _plan((X, Y, T, t_pad, Tp) + modes, z.device, _real_of(z.dtype))       # its inverse reconstructs Tp steps
    return hip_truncated_irfftn(z, plan, T, scale=fs, accumulate=accumulate, add_last=add_last,
                                kt_scale=_c2r_weights(modes[2], Tp, z.device, z.dtype, inverse=True))


def _inv_trunc_vjp(dy, cfg):
    """G^T(dy) for the zero-padded inverse transform G (cfg of ``_InvTruncFn``)."""
    (X, Y, T, t_pad, t_out, mx, my, mt), t_keep, norm = cfg
    _, is_ = _norm_scales(norm, X * Y * (T + t_pad), X * Y * t_out)
    gh, _ = hip_truncated_rfftn(dy.contiguous(), (mx, my, mt), t_pad=t_out - t_keep, t_out=t_out, scale=is_,
                                kt_scale=_c2r_weights(mt, t_out, dy.device, dy.dtype))
    return gh


def _contract_vjp(gh, vh, params, cfg, need_v, need_params):
    """(gv, parameter gradients) of the 4-corner contraction: the contraction kernels reading the weight blocks as their
    conjugate transpose (``tcfd_fno_contract_adjoint``), ONE launch of batch-summed outer products for all the weight and
    bias blocks (``tcfd_fno_contract_wgrad``), written in the parameters' own layout."""
    delta, modes, use_mfma, has_bias = cfg
    weights = params[:4]
    mx, my, mt = modes
    b, ci = vh.shape[:2]
    co = weights[0].shape[1]
    real = _real_of(vh.dtype)
    cplx = torch.complex128 if real == torch.float64 else torch.complex64
    code = _lib.TCFD_C128 if real == torch.float64 else _lib.TCFD_C64
    gh = gh.to(cplx).contiguous()
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream(vh.device).cuda_stream)
    grads = [None] * len(params)
    want_w = [bool(need_params[k]) for k in range(4)]
    want_b = [bool(has_bias and need_params[4 + k]) for k in range(4)]
    if any(want_w) or any(want_b):
        gw = [torch.empty(ci, co, mx, my, mt, dtype=cplx, device=vh.device) if w_ else None for w_ in want_w]
        gb = [torch.empty(mx, my, mt, dtype=cplx, device=vh.device) if b_ else None for b_ in want_b]
        arr = lambda ts: (ctypes.c_void_p * 4)(*[t.data_ptr() if t is not None else None for t in ts])
        with torch.cuda.device(vh.device):
            rc = lib.tcfd_fno_contract_wgrad(vh.data_ptr(), gh.data_ptr(), arr(gw), arr(gb), float(delta), b, ci, co, mx, my,
                                             mt, code, stream)
        _lib.check(rc, "tcfd_fno_contract_wgrad")

        def like(g, prm):
            g = g if prm.is_complex() else torch.view_as_real(g)
            return g.to(prm.dtype).reshape(prm.shape)
        for k in range(4):
            if gw[k] is not None:
                grads[k] = like(gw[k], params[k])
            if gb[k] is not None:
                grads[4 + k] = like(gb[k], params[4 + k])
    gv = None
    if need_v:
        ws_ = [(torch.view_as_real(w) if w.is_complex() else w).detach().to(device=vh.device, dtype=real).contiguous()
               for w in weights]
        gv = torch.empty(b, ci, 2 * mx, 2 * my, mt, dtype=cplx, device=vh.device)
        with torch.cuda.device(vh.device):
            rc = lib.tcfd_fno_contract_adjoint(gh.data_ptr(), _ptr_array(ws_), gv.data_ptr(), b, co, ci, mx, my, mt,
                                               1 if use_mfma else 0, code, stream)
        _lib.check(rc, "tcfd_fno_contract_adjoint")
    return gv, grads


class _FwdTruncFn(torch.autograd.Function):
    """F: truncated rfftn of the left-padded input on the HIP kernels.  For a complex cotangent z (torch's convention
    dL/dRe + i dL/dIm),  F^T(z) = G'(z / c_in): the zero-padded inverse of the plan with output length T_in + t_pad
    (kept tail T_in) applied to z with the interior time modes halved -- the c2r kernel doubles them, the adjoint of
    an r2c transform does not (c = 1 for DC / Nyquist in t, else 2)."""

    @staticmethod
    def forward(ctx, v, modes, t_pad, t_out, norm):
        vh, _ = hip_truncated_rfftn(v, modes, t_pad=t_pad, t_out=t_out, norm=norm)
        ctx.cfg = (tuple(v.shape), tuple(modes), t_pad, t_out, norm)
        return vh

    @staticmethod
    def backward(ctx, z):
        return _fwd_trunc_vjp(z, ctx.cfg), None, None, None, None


class _InvTruncFn(torch.autograd.Function):
    """G: zero-padded irfftn (c2r semantics, last t_keep of t_out steps).  G^T(dy) = c_out * F'(dy) with F' the forward
    transform of the plan whose input is t_keep steps left-padded to t_out."""

    @staticmethod
    def forward(ctx, oh, plan_key, t_keep, norm):
        plan = _plan(plan_key, oh.device, _real_of(oh.dtype))
        ctx.cfg = (plan_key, t_keep, norm)
        return hip_truncated_irfftn(oh, plan, t_keep, norm=norm)

    @staticmethod
    def backward(ctx, dy):
        return _inv_trunc_vjp(dy, ctx.cfg), None, None, None


class _ContractFn(torch.autograd.Function):
    """The 4-corner mode contraction (+ delta * bias) on the MFMA kernel; backward: ``_contract_vjp``."""

    @staticmethod
    def forward(ctx, vh, delta, modes, use_mfma, has_bias, *params):
        weights, bias = list(params[:4]), (list(params[4:8]) if has_bias else None)
        ctx.cfg = (delta, tuple(modes), use_mfma, has_bias)
        ctx.save_for_backward(vh, *params)
        return hip_contract(vh, [w.detach() for w in weights], [x.detach() for x in bias] if bias else None, delta, modes,
                            use_mfma=use_mfma)

    @staticmethod
    def backward(ctx, gh):
        vh, *params = ctx.saved_tensors
        gv, grads = _contract_vjp(gh, vh, params, ctx.cfg, ctx.needs_input_grad[0], ctx.needs_input_grad[5:])
        return (gv, None, None, None, None, *grads)


def hip_spectral_conv_autograd(v, weights, bias, delta, modes, t_pad, t_out, t_keep, norm, use_mfma=True, post=None):
    """Differentiable spectral convolution  y = G(post(W . F(v)))  (SURVEY 8f rank 4): three autograd Functions on the
    same HIP kernels as the forward-only path; ``post`` (e.g. the Helmholtz projection of SpectralConvT) acts on the
    small truncated spectrum with ordinary differentiable torch ops."""
    b, ci, X, Y, T = v.shape
    params = list(weights) + (list(bias) if bias is not None else [])
    vh = _FwdTruncFn.apply(v, tuple(modes), t_pad, t_out, norm)
    oh = _ContractFn.apply(vh, float(delta), tuple(modes), use_mfma, bias is not None, *params)
    if post is not None:
        oh = post(oh).to(vh.dtype)        # a float64 projection table promotes an fp32 layer's spectrum: back to the layer's
    return _InvTruncFn.apply(oh, (X, Y, T, t_pad, t_out) + tuple(modes), t_keep, norm)


# ----------------------------------------------------------------------------- pointwise helpers
class LayerNormnd(nn.GroupNorm):
    """GroupNorm with one group used as a LayerNorm over (C, *spatial)."""

    def __init__(self, num_channels, eps=1e-07, elementwise_affine=True, device=None, dtype=None):
        super().__init__(num_groups=1, num_channels=num_channels, eps=eps, affine=elementwise_affine,
                         device=device, dtype=dtype)


class PointwiseFFN(nn.Module):
    """Two 1x1 convolutions with a channel expansion and an activation in between."""

    def __init__(self, in_channels: int, out_channels: int, mid_channels: int, activation: ActivationType = "ReLU",
                 dim: int = 3):
        super().__init__()
        if dim not in conv_dict:
            raise ValueError(f"Unsupported dimension: {dim}, expected 1, 2, or 3")
        Conv = conv_dict[dim]
        self.linear1 = Conv(in_channels, mid_channels, 1)
        self.linear2 = Conv(mid_channels, out_channels, 1)
        self.activation = getattr(nn, activation)()

    def forward(self, v):
        out = hip_pointwise(v, self.linear1, self.activation, self.linear2)
        return out if out is not None else self.linear2(self.activation(self.linear1(v)))


def _pointwise_reference(spec, x, skip, w1, b1, w2, b2, ws, bs, gamma, beta):
    """The fused block written with channel einsums (differentiable torch ops) -- used for the backward recompute."""
    has_l1, act1, act2, mode, eps = spec
    b, ci = x.shape[:2]
    tail = x.shape[2:]
    h = x if eps is None else F.group_norm(x, 1, gamma, beta, eps)
    h = h.reshape(b, ci, -1)
    bias = lambda t: 0 if t is None else t[None, :, None]
    if has_l1:
        h = torch.einsum("mc,bcp->bmp", w1.reshape(w1.shape[0], -1), h) + bias(b1)
        h = act1(h) if act1 is not None else h
    o = torch.einsum("om,bmp->bop", w2.reshape(w2.shape[0], -1), h) + bias(b2)
    if mode == 1:
        o = o + torch.einsum("oc,bcp->bop", ws.reshape(ws.shape[0], -1), skip.reshape(b, skip.shape[1], -1)) + bias(bs)
    o = o.reshape(b, o.shape[1], *tail)
    if mode == 2:
        o = o + skip[..., -1:]
    return act2(o) if act2 is not None else o


_BWD_LAYOUTS: Dict[tuple, Optional[tuple]] = {}


def _pointwise_bwd_layout(has_l1, b, ci, cm, co, P, T, sT, c1, c2, mode):
    """Layout query of ``tcfd_fno_pointwise_bwd`` (no data): the six geometry numbers, or None when the backward kernel is
    not instantiated for the combination.  Remembered per (device, shape, switches): the query asks the runtime for an
    occupancy figure, ~40 us that a small-batch training iteration pays six times."""
    key = (torch.cuda.current_device(), has_l1, b, ci, cm, co, P, T, sT, c1, c2, mode,
           os.environ.get("TCFD_PW_BWD_TILES"), os.environ.get("TCFD_PWB_REDUCE1"))
    if key in _BWD_LAYOUTS:
        hit = _BWD_LAYOUTS[key]
        return (ctypes.c_int * 6)(*hit) if hit is not None else None
    dims = (ctypes.c_int * 6)()
    one = ctypes.c_void_p(1) if has_l1 else None   # only null / non-null of w1 matters
    rc = _lib.load().tcfd_fno_pointwise_bwd(None, None, None, None, None, one, None, None, None, None, None, None, 0, dims, b, ci,
                                            cm, co, P, T, sT, c1, c2, mode, 0, None)
    _BWD_LAYOUTS[key] = tuple(dims) if rc == 0 else None
    return dims if rc == 0 else None


def _sum_rows(mat: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """Column sums (float64) of the first ``rows * cols`` values of an fp32 buffer read as a (rows, cols) matrix
    (``tcfd_sum_rows``: the per-wave rows of partial sums of the backward kernels)."""
    lib = _lib.load()
    out = torch.empty(cols, dtype=torch.float64, device=mat.device)
    scratch = torch.empty(lib.tcfd_sum_rows_slices(rows) * cols, dtype=torch.float64, device=mat.device)
    with torch.cuda.device(mat.device):
        _lib.check(lib.tcfd_sum_rows(mat.data_ptr(), out.data_ptr(), scratch.data_ptr(), rows, cols,
                                     ctypes.c_void_p(torch.cuda.current_stream(mat.device).cuda_stream)), "tcfd_sum_rows")
    return out


def _hip_pointwise_backward(spec, dout, x, skip, w1, b1, w2, b2, ws, bs, gamma, beta, need_dx: bool = True, out=None,
                            compact_skip: bool = False):
    """Gradients of the fused block from ``tcfd_fno_pointwise_bwd`` (one pass; weight gradients accumulated on MFMA,
    per-wave partial sums added here).  None when the combination is not covered: a folded LayerNorm or a width
    that is not instantiated -- the caller then recomputes the
    block with torch einsums.  ``out``: what the forward kept for this call (``_saved_kind``): the block's forward output
    (ReLU: the kernel reads the output mask from it) or its pre-activation (other activations), handed to
    ``tcfd_fno_pointwise_bwd_out`` -- nothing of the pre-activation is recomputed then."""
    has_l1, act1, act2, mode, eps = spec
    c1, c2 = _act_code(act1), _act_code(act2)
    if c1 is None or c2 is None or not x.is_cuda or x.dtype != torch.float32:
        return None
    if eps is not None:
        return (_hip_norm_proj_backward(eps, dout, x, w2, b2, gamma, beta, need_dx=need_dx)
                if (not has_l1 and mode == 0 and c2 == 0) else None)
    b, ci = x.shape[:2]
    co = w2.shape[0]
    cm = w1.shape[0] if has_l1 else ci
    P = x[0, 0].numel()
    lib = _lib.load()
    T = x.shape[-1]
    sT = skip.shape[-1] if mode == 2 else 0
    # the lifting tail's compact skip gradient can leave the kernel already summed over t (skip_mode 3: whole rows of T steps inside
    # 1 or 5 consecutive groups of 16 points) -- no (b, co, P) tensor of dL/dz2 and no pass that adds it up
    kept_ok = (out is not None and out.dtype == torch.float32 and out.device == x.device and out.numel() == dout.numel()
               and out.shape[:2] == dout.shape[:2])     # (what the call below hands to the kernel as the kept tensor)
    tsum = (mode == 2 and compact_skip and (kept_ok or c2 == 0) and os.environ.get("TCFD_PWB_TSUM", "1") != "0")
    dims = _pointwise_bwd_layout(has_l1, b, ci, cm, co, P, T, sT, c1, c2, 3) if tsum else None
    tsum = dims is not None
    if dims is None:
        dims = _pointwise_bwd_layout(has_l1, b, ci, cm, co, P, T, sT, c1, c2, mode)
    if dims is None:
        return None
    queried = tuple(dims)
    COP, CB, CM1, CIP, per_row, _ = queried
    dev = x.device
    xs, dz = x.detach().contiguous(), dout.detach().contiguous()
    sk = skip.detach().contiguous() if mode else None
    mat = lambda w: w.detach().reshape(w.shape[0], -1)
    w1m = mat(w1).contiguous() if has_l1 else None
    w2t = mat(w2).t().contiguous()
    wst = mat(ws).t().contiguous() if mode == 1 else None
    vec = lambda t: t.detach().contiguous() if t is not None else None
    b1v, b2v, bsv = vec(b1), vec(b2), vec(bs)
    dx = torch.empty_like(xs) if need_dx else None
    if tsum:
        ds = torch.empty(*skip.shape[:-1], 1, dtype=torch.float32, device=dev)       # (b, co, X, Y, 1): the t-sums themselves
    else:
        ds = torch.empty_like(sk) if mode == 1 else (torch.empty(b, co, *x.shape[2:], dtype=torch.float32, device=dev)
                                                     if mode == 2 else None)
    max_waves = max(2048, int(dims[5]))     # (the layout query of the tiled kernel reports the rows of a launch that fills the device)
    # zeros: the all-MFMA kernel writes only the entries of a row that mean something (one row per wave)
    partials = torch.zeros(max_waves, per_row, dtype=torch.float32, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None
    ys = None
    if (out is not None and out.dtype == torch.float32 and out.device == dev and out.numel() == dz.numel()
            and out.shape[:2] == dz.shape[:2]):
        ys = out.detach().contiguous()
    with torch.cuda.device(dev):
        rc = lib.tcfd_fno_pointwise_bwd_out(ptr(xs), ptr(sk), ptr(dz), ptr(ys), ptr(dx), ptr(ds), ptr(w1m), ptr(b1v), ptr(w2t),
                                            ptr(b2v), ptr(wst), ptr(bsv), ptr(partials), max_waves, dims, b, ci, cm, co, P, T, sT, c1,
                                            c2, 3 if tsum else mode, 0, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(rc, "tcfd_fno_pointwise_bwd_out")
    if tuple(dims)[:5] != queried[:5]:
        # the launch rewrites `dims` with the geometry of the kernel it actually ran (entry 5, the rows a launch fills, is
        # not part of the row layout: the LDS-staged kernel reports it only from a launch); the buffers above were sized and
        # the scatter below is laid out from the data-less query.  They differ only if the launch declined the kernel the query
        # chose (a kept tensor it rejects, a switch flipped between the forward and the backward): the partial sums then have
        # another row layout and every weight gradient would be silently wrong.
        raise _lib.TcfdError(f"tcfd_fno_pointwise_bwd_out ran with row layout {tuple(dims)}, the layout query said {queried}")
    # the summed row is A (COP x CB) = [dW2 | db2 | dWs] followed by B (CM1 x CIP) = [dW1 | db1]: every gradient leaves the final
    # pass of the row sum as a dense tensor of its parameter's shape (tcfd_sum_rows_scatter)
    ch = cm if has_l1 else ci
    new = lambda like: torch.empty(like.shape, dtype=torch.float32, device=dev)
    g_w2, g_b2 = new(w2), (new(b2) if b2 is not None else None)
    g_ws = new(ws) if mode == 1 else None
    g_bs = new(bs) if (mode == 1 and bs is not None) else None
    g_w1 = new(w1) if has_l1 else None
    g_b1 = new(b1) if (has_l1 and b1 is not None) else None
    segs = [(g_w2, 0, co, ch, CB), (g_b2, ch, co, 1, CB), (g_ws, ch + 1, co, ci, CB), (g_bs, ch, co, 1, CB),
            (g_w1, COP * CB, cm, ci, CIP), (g_b1, COP * CB + ci, cm, 1, CIP)]
    segs = [sg for sg in segs if sg[0] is not None]
    table = (ctypes.c_long * (4 * len(segs)))(*[v for sg in segs for v in sg[1:]])
    dsts = (ctypes.c_void_p * len(segs))(*[sg[0].data_ptr() for sg in segs])
    scratch = torch.empty(lib.tcfd_sum_rows_slices(int(dims[5])) * per_row, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.tcfd_sum_rows_scatter(partials.data_ptr(), scratch.data_ptr(), int(dims[5]), per_row, len(segs), table, dsts,
                                             ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "tcfd_sum_rows_scatter")
    if tsum:
        g_skip = ds.to(skip.dtype) if skip.dtype != ds.dtype else ds
    elif mode == 2:   # the skip's last time slice was broadcast over t: its gradient is the t-sum of dL/dz2
        # (compact_skip: the sums alone, (b, co, X, Y, 1) -- the caller joins them to the last step of another gradient itself)
        g_skip = (torch.empty(*skip.shape[:-1], 1, dtype=skip.dtype, device=dev) if compact_skip
                  else torch.empty_like(skip, memory_format=torch.contiguous_format))
        sT = 1 if compact_skip else sT
        with torch.cuda.device(dev):
            _lib.check(lib.tcfd_sum_t_into_last(ds.data_ptr(), g_skip.data_ptr(), ds.numel() // T, T, sT,
                                                ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "tcfd_sum_t_into_last")
    else:
        g_skip = ds.view_as(skip) if ds is not None else None
    return (dx.view_as(x) if dx is not None else None, g_skip, g_w1, g_b1, g_w2, g_b2, g_ws, g_bs, None, None)


def _hip_norm_proj_backward(eps, dout, x, w, bias, gamma, beta, need_dx: bool = True, moments=None, table=None):
    """Backward of ``proj(LayerNorm(x))`` (one group over (C, X, Y, T), per-channel affine) in two passes over the data.

    Pass 1 (``tcfd_fno_pointwise_bwd`` with per-sample partial sums, no dx): M_b[o, c] = sum_p dy[o] x[c] and
    M_b[o, C] = sum_p dy[o] on MFMA.  Everything the LayerNorm backward needs follows from these (b, Co, C+1) numbers and
    the per-sample statistics: with x^ = (x - mu) r and g = W^T dy,
        sum_p dy[o] x^[c] = r (M[o,c] - mu M[o,C]),   dW = sum_b gamma (.) + beta M[.,C],   dbias = sum_b M[.,C],
        dgamma_c = sum_b sum_o W[o,c] sum_p dy[o] x^[c],   dbeta_c = sum_b sum_o W[o,c] M[o,C],
        A_b = sum_{c,p} gamma_c g,   B_b = sum_{c,p} gamma_c g x^.
    Pass 2: dx = r (gamma g - A/L - x^ B/L) = (per-sample 1x1x1 convolution of dy) + alpha_b + kappa_b x  -- the
    convolution on ``tcfd_fno_pointwise`` with per-sample weights, the rank-one correction as one fused multiply-add."""
    # ``table`` (C, P): x is the ONE-channel input (b, P) and the block input is x + table[c] (the lifting operator): the
    # weight-gradient pass reads them through the kernel's ``pe`` mode; needs ``moments`` and ``need_dx=False``
    b = x.shape[0]
    C = table.shape[0] if table is not None else x.shape[1]
    co = w.shape[0]
    P = table.shape[1] if table is not None else x[0, 0].numel()
    L = C * P
    dev = x.device
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    if table is not None and (moments is None or need_dx):
        raise ValueError("the table form needs the forward's moments and does not produce an input gradient")
    dims = (ctypes.c_int * 6)()
    if lib.tcfd_fno_pointwise_bwd(None, None, None, None, None, None, None, None, None, None, None, None, 0, dims, b, C, C, co,
                                  P, 0, 0, 0, 0, 0, 1, None) != 0:
        return None
    COP, CB, _, _, per_row, _ = list(dims)
    xs, dz = x.detach().contiguous(), dout.detach().contiguous()
    stats = moments if moments is not None else torch.empty(b, 2, dtype=torch.float64, device=dev)   # (sum, sum of squares) per sample
    W = w.detach().reshape(co, C)
    w2t = W.t().contiguous()
    if C <= 47 and co <= 32 and P % 16 == 0 and os.environ.get("TCFD_OUTER_SUMS_MFMA", "1") != "0":
        # the sums on MFMA straight from two 16-byte loads per lane (tcfd_fno_sample_outer_sums)
        wps = int(os.environ.get("TCFD_OUTER_WPS", 128))    # 64: 0.49 ms, 128: 0.30, 256: 0.33 at config 5 (tests/micro/outer_sums_timing.py)
        R16, C16 = 16 * ((co + 15) // 16), 16 * ((C + 16) // 16)      # the sums come as (R16 x C16) matrices of 16 x 16 tiles
        tiles = torch.empty(wps, b, R16 * C16, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if moments is None:
                _lib.check(lib.tcfd_row_moments(xs.data_ptr(), stats.data_ptr(), b, L, stream), "tcfd_row_moments")
            _lib.check(lib.tcfd_fno_sample_outer_sums(dz.data_ptr(), xs.data_ptr(), table.data_ptr() if table is not None else None,
                                                      tiles.data_ptr(), b, C, co, P, wps, stream), "tcfd_fno_sample_outer_sums")
        M = _sum_rows(tiles, wps, b * R16 * C16).view(b, R16, C16)
        return _norm_proj_grads(eps, dz, xs, W, bias, gamma, beta, M[:, :co, :C], M[:, :co, C], stats, need_dx, w, x, L, P, stream)
    max_waves = 2048 + b
    partials = torch.empty(max_waves, per_row, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if moments is None:
            _lib.check(lib.tcfd_row_moments(xs.data_ptr(), stats.data_ptr(), b, L, stream), "tcfd_row_moments")
        if table is not None:
            rc = lib.tcfd_fno_pointwise_bwd_pe(xs.data_ptr(), table.data_ptr(), dz.data_ptr(), None, w2t.data_ptr(), None,
                                               partials.data_ptr(), max_waves, dims, b, C, co, P, 1, stream)
        else:
            rc = lib.tcfd_fno_pointwise_bwd(xs.data_ptr(), None, dz.data_ptr(), None, None, None, None, w2t.data_ptr(), None, None,
                                            None, partials.data_ptr(), max_waves, dims, b, C, C, co, P, 0, 0, 0, 0, 0, 1, stream)
    _lib.check(rc, "tcfd_fno_pointwise_bwd")
    rows = dims[5]
    M = _sum_rows(partials, rows // b, b * per_row).view(b, per_row)[:, : COP * CB].view(b, COP, CB)
    Mx, M1 = M[:, :co, :C], M[:, :co, C]                                   # (b, co, C), (b, co)
    return _norm_proj_grads(eps, dz, xs, W, bias, gamma, beta, Mx, M1, stats, need_dx, w, x, L, P, stream)


def _norm_proj_grads(eps, dz, xs, W, bias, gamma, beta, Mx, M1, stats, need_dx, w, x, L, P, stream):
    """Second half of ``_hip_norm_proj_backward``: everything after the per-sample sums Mx (b, co, C), M1 (b, co)."""
    b, co, C = Mx.shape
    dev = dz.device
    lib = _lib.load()
    mu = stats[:, 0] / L
    r = torch.rsqrt((stats[:, 1] / L - mu * mu).clamp_min(0) + eps)
    Wd = W.double()
    g_ = gamma.detach().double() if gamma is not None else torch.ones(C, dtype=torch.float64, device=dev)
    b_ = beta.detach().double() if beta is not None else torch.zeros(C, dtype=torch.float64, device=dev)
    Sx = r[:, None, None] * (Mx - mu[:, None, None] * M1[:, :, None])      # sum_p dy[o] x^[c]
    g_w = (g_[None, None, :] * Sx + b_[None, None, :] * M1[:, :, None]).sum(0)
    g_b = M1.sum(0)
    # (broadcast products + sums, not einsum: these few hundred numbers would otherwise go through a vendor GEMM kernel, the
    #  only library kernel in a config-5 training trace)
    G1 = (Wd[None] * M1[:, :, None]).sum(1)                                 # sum_p g[c]        = sum_o W[o,c] M1[b,o]
    Gx = (Wd[None] * Sx).sum(1)                                             # sum_p g[c] x^[c]  = sum_o W[o,c] Sx[b,o,c]
    g_beta, g_gamma = G1.sum(0), Gx.sum(0)
    cast = lambda t, like: t.to(like.dtype).reshape(like.shape) if like is not None else None
    if not need_dx:      # the block's input is data (e.g. the network input + positional encoding): pass 2 is not needed
        return (None, None, None, None, cast(g_w, w), cast(g_b, bias), None, None, cast(g_gamma, gamma), cast(g_beta, beta))
    A = (g_[None] * G1).sum(1)
    B = (g_[None] * Gx).sum(1)
    # pass 2: dx = sum_o (r gamma_c W[o,c]) dy[o] + alpha_b + kappa_b x
    wb = (r[:, None, None] * g_[None, None, :] * Wd[None]).float().contiguous()          # (b, co, C) = w2t per sample
    kappa = -(r * r) * B / L
    alpha = (-r * A / L - kappa * mu).float()
    bias2 = alpha[:, None].expand(b, C).contiguous()
    dx = torch.empty_like(xs)
    with torch.cuda.device(dev):
        rc = lib.tcfd_fno_pointwise(dz.data_ptr(), None, dx.data_ptr(), None, None, wb.data_ptr(), bias2.data_ptr(), None, None,
                                    b, co, co, C, P, x.shape[-1], 0, 0, 0, 0, co * C, C, None, stream)
    if rc == -1 and b"not instantiated" in lib.tcfd_last_error():
        return None
    _lib.check(rc, "tcfd_fno_pointwise")
    dx.addcmul_(xs, kappa.float().view(b, *([1] * (x.dim() - 1))))
    return (dx.view_as(x), None, None, None, cast(g_w, w), cast(g_b, bias), None, None, cast(g_gamma, gamma), cast(g_beta, beta))


def _saved_kind(spec, ci: int, cm: int, co: int, P: int) -> int:
    """What the backward kernel of a fused block wants the forward to keep (``tcfd_fno_pointwise_bwd_saved``): 0 nothing,
    1 the block's OUTPUT (ReLU output activation: its sign is the mask; it is the next layer's input anyway, so it costs no
    memory), 2 the PRE-activation of the output activation (GELU / SiLU / tanh on the tiled all-MFMA kernel: one more
    activation-sized tensor per block instead of recomputing W2.h + Ws.skip in the backward)."""
    has_l1, act1, act2, _, eps = spec
    c1, c2 = _act_code(act1), _act_code(act2)
    if not has_l1 or eps is not None or c1 is None or c2 is None:
        return 0
    return int(_lib.load().tcfd_fno_pointwise_bwd_saved(int(ci), int(cm), int(co), int(P), c1, c2))


class _PointwiseFn(torch.autograd.Function):
    """Fused pointwise block under autograd: the forward value comes from the HIP kernel (passed in), the backward
    recomputes the block from its inputs (nothing but the inputs is kept alive between forward and backward)."""

    @staticmethod
    def forward(ctx, out, kept, spec, *tensors):
        ctx.spec = spec
        ctx.present = [t is not None for t in tensors]
        # ``kept`` (see _saved_kind): the block's output (ReLU: its sign is the mask; it IS the next block's input, no extra
        # memory) or its pre-activation (other activations), or None
        ctx.keeps = kept is not None
        ctx.save_for_backward(*([kept] if ctx.keeps else []), *[t for t in tensors if t is not None])
        return out.view_as(out)

    @staticmethod
    def backward(ctx, dout):
        saved = list(ctx.saved_tensors)
        y = saved.pop(0) if ctx.keeps else None
        it = iter(saved)
        tensors = [next(it) if p else None for p in ctx.present]
        need = ctx.needs_input_grad[3:]
        hip = _hip_pointwise_backward(ctx.spec, dout, *tensors, need_dx=bool(need[0]), out=y)
        if hip is not None:
            return (None, None, None, *[g if n else None for g, n in zip(hip, need)])
        with torch.enable_grad():
            leaves = [t.detach().requires_grad_(True) if (t is not None and n) else (t.detach() if t is not None else None)
                      for t, n in zip(tensors, need)]
            out = _pointwise_reference(ctx.spec, *leaves)
            wanted = [l for l, n in zip(leaves, need) if l is not None and n]
            got = iter(torch.autograd.grad(out, wanted, dout, allow_unused=True))
        grads = [next(got) if (l is not None and n) else None for l, n in zip(leaves, need)]
        return (None, None, None, *grads)


class _SpectralLayerFn(torch.autograd.Function):
    """One autograd node for a whole spectral layer  out = act2(FFN(conv(v)) + skip(v)):  the forward values come from the
    HIP kernels (passed in), the backward chains the same pieces as the separate nodes (``_hip_pointwise_backward``,
    ``_inv_trunc_vjp``, ``_contract_vjp``, ``_fwd_trunc_vjp``) -- but as one node it knows that the layer input feeds BOTH
    the convolution and the skip path, so the last inverse transform ADDS its result to the skip gradient in its own store
    loop (``tcfd_fno_inverse_trunc_acc``) instead of autograd adding two activation-sized tensors (0.42 ms per layer at
    config 5)."""

    @staticmethod
    def forward(ctx, out, kept, x1, vh, cfg, v, *params):
        ctx.cfg = cfg
        ctx.present = [t is not None for t in params]
        ctx.keeps = kept is not None               # the layer output (ReLU mask) or the pre-activation, see _saved_kind
        ctx.save_for_backward(*([kept] if ctx.keeps else []), x1, vh, v, *[t for t in params if t is not None])
        return out.view_as(out)

    @staticmethod
    def backward(ctx, dout):
        spec, n_conv, ccfg, fwd_cfg, inv_cfg = ctx.cfg
        saved = list(ctx.saved_tensors)
        y = saved.pop(0) if ctx.keeps else None
        x1, vh, v, *rest = saved
        it = iter(rest)
        params = [next(it) if p else None for p in ctx.present]
        conv_params, pw = params[:n_conv], params[n_conv:]
        need = ctx.needs_input_grad[6:]
        need_v = ctx.needs_input_grad[5]
        # the lifting tail (skip = last time slice of v): its t-summed gradient stays compact and joins the last step of the
        # transform's adjoint in that transform's store loop -- no zero-filled (b, C, X, Y, T) tensor written and read back
        compact = spec[3] == 2 and need_v and os.environ.get("TCFD_COMPACT_SKIP_GRAD", "1") != "0"
        hip = _hip_pointwise_backward(spec, dout, x1, v if spec[3] else None, *pw, None, None, out=y, compact_skip=compact)
        if hip is None:
            raise _lib.TcfdError("pointwise backward kernel not available for a layer that was admitted to the fused path")
        dx1, g_skip = hip[0], hip[1]
        gh = _inv_trunc_vjp(dx1, inv_cfg)
        gv, cgrads = _contract_vjp(gh, vh, conv_params, ccfg, need_v, list(need[:n_conv]) + [False] * 8)
        dv = None
        if need_v and compact:
            dv = _fwd_trunc_vjp(gv, fwd_cfg, add_last=g_skip)
        elif need_v:
            acc = g_skip if (g_skip is not None and g_skip.is_contiguous() and g_skip.shape == v.shape) else None
            dv = _fwd_trunc_vjp(gv, fwd_cfg, accumulate=acc)
            if acc is None and g_skip is not None:
                dv = dv + g_skip
        pgrads = [g if n else None for g, n in zip(hip[2:8], need[n_conv:])]
        cgrads = [g if n else None for g, n in zip(cgrads, need[:n_conv])]
        return (None, None, None, None, None, dv, *cgrads, *pgrads)


def hip_spectral_layer(conv, v, lin1, act1, lin2, skip_conv=None, act2=None, skip_last_slice: bool = False,
                       out_steps: Optional[int] = None) -> Optional[torch.Tensor]:
    """Training form of one SFNO layer, ``act2(FFN(conv(v)) + skip_conv(v))`` (fno/sfno.py:607-614) or of the lifting tail
    ``act2(v[..., -1:] + FFN(conv(v)))`` (:258-259), as ONE autograd node (``_SpectralLayerFn``).  Returns None when gradients
    are not being recorded or the combination is not covered; the caller then composes ``conv(v)`` and ``hip_pointwise``."""
    if (not torch.is_grad_enabled() or not v.is_cuda or v.dtype != torch.float32 or v.dim() != 5
            or os.environ.get("TCFD_FUSED_LAYER_GRAD", "1") == "0" or not hasattr(conv, "_plain_args")
            or not _fused_xy(v.shape[2], v.shape[3])):
        return None
    # a user subclass that overrides forward() / spectral_conv() is a different convolution: it must run under grad exactly
    # as it runs under no_grad (through conv(v)), never be rebuilt from the parent's parameters
    cls = type(conv)
    if cls.forward not in (SpectralConvS.forward, SpectralConvT.forward) or cls.spectral_conv is not SpectralConvS.spectral_conv \
            or cls._plain_args not in (SpectralConvS._plain_args, SpectralConvT._plain_args):
        return None
    cargs = conv._plain_args(v, out_steps)
    if cargs is None:
        return None
    weights, bias, delta, modes, t_pad, t_out, t_keep, norm = cargs
    if not _library_takes(v.shape[2], v.shape[3], v.shape[4], t_pad, t_out, modes, t_keep, v.device):
        return None          # (the composed path: conv(v) then differentiates through the dense transforms)
    conv_params = list(weights) + (list(bias) if bias is not None else [])
    pw = lambda m, a: getattr(m, a) if m is not None else None
    pw_t = [pw(lin1, "weight"), pw(lin1, "bias"), lin2.weight, lin2.bias, pw(skip_conv, "weight"), pw(skip_conv, "bias")]
    every = [v] + conv_params + [t for t in pw_t if t is not None]
    if not any(t.requires_grad for t in every) or any(t.dtype != torch.float32 for t in every if not t.is_complex()):
        return None
    if any(t.is_complex() and t.dtype != torch.complex64 for t in every) or v.shape[0] == 0:
        return None
    c1, c2 = _act_code(act1), _act_code(act2)
    mode = 1 if skip_conv is not None else (2 if skip_last_slice else 0)
    b, ci, X, Y, T = v.shape
    co_conv = weights[0].shape[1]
    if c1 is None or c2 is None or not _is_pointwise(lin2) or (lin1 is not None and not _is_pointwise(lin1)):
        return None
    cm = lin1.out_channels if lin1 is not None else co_conv
    if _pointwise_bwd_layout(lin1 is not None, b, co_conv, cm, lin2.out_channels, X * Y * t_keep, t_keep, T if mode == 2 else 0,
                             c1, c2, mode) is None:
        return None
    spec = (lin1 is not None, act1, act2, mode, None)
    kind = _saved_kind(spec, co_conv, cm, lin2.out_channels, X * Y * t_keep)
    with torch.no_grad():
        vh, plan = hip_truncated_rfftn(v, modes, t_pad=t_pad, t_out=t_out, norm=norm)
        oh = hip_contract(vh, weights, bias, delta, modes)
        x1 = hip_truncated_irfftn(oh, plan, t_keep, norm=norm)
        z2 = torch.empty(b, lin2.out_channels, X, Y, t_keep, dtype=torch.float32, device=v.device) if kind == 2 else None
        out = hip_pointwise(x1, lin1, act1, lin2, skip=v if mode else None, skip_conv=skip_conv, act2=act2,
                            skip_last_slice=skip_last_slice, pre=z2)
    if out is None:
        return None
    cfg = (spec, len(conv_params), (float(delta), tuple(modes), True, bias is not None),
           (tuple(v.shape), tuple(modes), t_pad, t_out, norm), ((X, Y, T, t_pad, t_out) + tuple(modes), t_keep, norm))
    return _SpectralLayerFn.apply(out, out if kind == 1 else z2, x1, vh, cfg, v, *conv_params, *pw_t)


_ACT_CODES = {nn.Identity: 0, nn.ReLU: 1, nn.GELU: 2, nn.SiLU: 3, nn.Tanh: 4}


def _act_code(mod) -> Optional[int]:
    if mod is None:
        return 0
    code = _ACT_CODES.get(type(mod))
    if code == 2 and getattr(mod, "approximate", "none") != "none":
        return None
    return code


def _is_pointwise(conv) -> bool:
    return (isinstance(conv, (nn.Conv1d, nn.Conv2d, nn.Conv3d)) and all(k == 1 for k in conv.kernel_size)
            and conv.groups == 1 and all(s_ == 1 for s_ in conv.stride) and all(p_ == 0 for p_ in conv.padding))


def hip_pointwise(x: torch.Tensor, lin1, act1, lin2, skip=None, skip_conv=None, act2=None,
                  skip_last_slice: bool = False, norm: Optional[nn.GroupNorm] = None,
                  out: Optional[torch.Tensor] = None, pre: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """out = act2( lin2(act1(lin1(x))) [+ skip_conv(skip) | + skip[..., -1:]] ) in ONE fused HIP kernel
    (``tcfd_fno_pointwise``); ``lin1=None`` makes it a single 1x1x1 convolution.  Returns ``None`` when the
    combination is not covered (channel counts, activation type, dtype, autograd) -- the caller then runs its
    torch modules.  ``pre`` (forward only, float32, the output's shape): also receives the pre-activation of ``act2``
    (``tcfd_fno_pointwise_pre``: what the backward of a GELU / SiLU / tanh block reads instead of recomputing it)."""
    c1, c2 = _act_code(act1), _act_code(act2)
    if (c1 is None or c2 is None or not x.is_cuda or x.dtype not in (torch.float32, torch.float64) or not _is_pointwise(lin2)
            or (lin1 is not None and not _is_pointwise(lin1)) or (skip_conv is not None and not _is_pointwise(skip_conv))):
        return None
    real = x.dtype
    for mod in (lin1, lin2, skip_conv, norm):
        for prm in (mod.parameters() if mod is not None else ()):
            if prm.dtype != real or prm.device != x.device:   # torch's own conv raises on this too
                raise TypeError(f"{real} input on {x.device} but a {prm.dtype} parameter on {prm.device}: build / move the "
                                "layer in the input's precision on the input's device")
    if skip is not None and (skip.dtype != real or skip.device != x.device):
        raise TypeError("the skip input must have the precision and the device of x")
    if real == torch.float64:
        return _hip_pointwise_f64(x, lin1, c1, lin2, skip, skip_conv, c2, skip_last_slice, norm)
    if torch.is_grad_enabled():
        mods = [m for m in (lin1, lin2, skip_conv, norm) if m is not None]
        tensors = [x, skip] + [p for m in mods for p in m.parameters()]
        if any(t is not None and t.requires_grad for t in tensors):
            # training: HIP forward, backward by recomputing the block with channel einsums under autograd
            # (torch's own Conv3d 1x1x1 backward takes SECONDS per layer at this size on ROCm)
            spec = (lin1 is not None, act1, act2, 1 if skip_conv is not None else (2 if skip_last_slice else 0),
                    norm.eps if norm is not None else None)
            kind = _saved_kind(spec, x.shape[1], lin1.out_channels if lin1 is not None else x.shape[1], lin2.out_channels,
                               x[0, 0].numel())
            with torch.no_grad():
                z2 = (torch.empty(x.shape[0], lin2.out_channels, *x.shape[2:], dtype=torch.float32, device=x.device)
                      if kind == 2 else None)
                out = hip_pointwise(x, lin1, act1, lin2, skip=skip, skip_conv=skip_conv, act2=act2,
                                    skip_last_slice=skip_last_slice, norm=norm, pre=z2)
            if out is None:
                return None
            pw = lambda m, a: getattr(m, a) if m is not None else None
            return _PointwiseFn.apply(out, out if kind == 1 else z2, spec, x, skip, pw(lin1, "weight"), pw(lin1, "bias"),
                                      lin2.weight, lin2.bias, pw(skip_conv, "weight"), pw(skip_conv, "bias"), pw(norm, "weight"),
                                      pw(norm, "bias"))
    b, ci = x.shape[:2]
    co = lin2.out_channels
    cm = lin1.out_channels if lin1 is not None else ci
    if lin2.in_channels != cm or (lin1 is not None and lin1.in_channels != ci):
        return None
    P = x[0, 0].numel()
    T = x.shape[-1]
    mode, s_t, sT = 0, None, 0
    if skip_conv is not None:
        if skip is None or skip.shape != x.shape or skip_conv.in_channels != ci or skip_conv.out_channels != co:
            return None
        mode, s_t = 1, skip.contiguous()
    elif skip_last_slice:
        if skip is None or skip.shape[1] != co or skip.shape[2:-1] != x.shape[2:-1]:
            return None
        mode, s_t, sT = 2, skip.contiguous(), skip.shape[-1]
    x = x.contiguous()
    if out is None:
        out = torch.empty(b, co, *x.shape[2:], dtype=torch.float32, device=x.device)
    elif (tuple(out.shape) != (b, co) + tuple(x.shape[2:]) or out.dtype != torch.float32 or out.device != x.device
          or not out.is_contiguous() or (torch.is_grad_enabled() and out.requires_grad)):
        raise ValueError("out must be a contiguous float32 tensor of the block's output shape on x's device (forward only)")
    if pre is not None and (pre.shape != out.shape or pre.dtype != torch.float32 or pre.device != x.device
                            or not pre.is_contiguous() or norm is not None):
        raise ValueError("pre must be a contiguous float32 tensor of the block's output shape on x's device (no folded LayerNorm)")

    def mat(conv, transpose):
        w = conv.weight.detach().reshape(conv.out_channels, conv.in_channels)
        return (w.t() if transpose else w).contiguous()

    w1 = mat(lin1, False) if lin1 is not None else None
    w2t = mat(lin2, True)
    w2_bs = b2_bs = 0
    folded_bias = None
    if norm is not None:
        # LayerNormnd (GroupNorm, one group) followed by a 1x1x1 convolution: the statistics come from the
        # many-workgroup moments kernel, the normalisation + affine is folded into per-sample weights
        #   W'_b[c, o] = W[o, c] gamma_c rstd_b ,  b'_b[o] = bias[o] + sum_c W[o, c] (beta_c - gamma_c mu_b rstd_b)
        if lin1 is not None or norm.num_groups != 1 or norm.num_channels != ci:
            return None
        L = ci * P
        stats = torch.empty(b, 2, dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().tcfd_row_moments(x.data_ptr(), stats.data_ptr(), b, L, ctypes.c_void_p(
                torch.cuda.current_stream(x.device).cuda_stream)), "tcfd_row_moments")
        mu = stats[:, 0] / L
        rstd = torch.rsqrt((stats[:, 1] / L - mu * mu).clamp_min(0) + norm.eps)
        gamma = norm.weight.detach().double() if norm.weight is not None else torch.ones(ci, dtype=torch.float64, device=x.device)
        beta = norm.bias.detach().double() if norm.bias is not None else torch.zeros(ci, dtype=torch.float64, device=x.device)
        W = lin2.weight.detach().reshape(co, ci).double()
        w2t = (W.t()[None] * (gamma[None, :, None] * rstd[:, None, None])).float().contiguous()      # (b, ci, co)
        shift = beta[None, :] - gamma[None, :] * (mu * rstd)[:, None]                               # (b, ci)
        folded_bias = (shift[:, None, :] * W[None]).sum(-1)        # (b, co); a broadcast sum: no vendor GEMM for b x ci x co numbers
        if lin2.bias is not None:
            folded_bias = folded_bias + lin2.bias.detach().double()[None]
        folded_bias = folded_bias.float().contiguous()                                              # (b, co)
        w2_bs, b2_bs = ci * co, co
    wst = mat(skip_conv, True) if skip_conv is not None else None
    ptr = lambda t: t.data_ptr() if t is not None else None
    bias = lambda c: c.bias.detach().contiguous() if (c is not None and c.bias is not None) else None
    b1, b2, bs = bias(lin1), bias(lin2), bias(skip_conv)
    if folded_bias is not None:
        b2 = folded_bias
    lib = _lib.load()
    with torch.cuda.device(x.device):
        rc = lib.tcfd_fno_pointwise_pre(x.data_ptr(), ptr(s_t), out.data_ptr(), ptr(pre), ptr(w1), ptr(b1), ptr(w2t), ptr(b2),
                                        ptr(wst), ptr(bs), b, ci, cm, co, P, T, sT, c1, c2, mode, w2_bs, b2_bs, None,
                                        ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc == -1 and b"not instantiated" in lib.tcfd_last_error():
        _note_torch_modules(f"pointwise block {ci} -> {cm} -> {co}")
        return None
    _lib.check(rc, "tcfd_fno_pointwise")
    return out


def _hip_pointwise_f64(x, lin1, c1, lin2, skip, skip_conv, c2, skip_last_slice, norm=None) -> Optional[torch.Tensor]:
    """The block in float64 (``tcfd_fno_pointwise_f64``); forward only -- under autograd (or for a width that is not
    instantiated) ``None``: the layer's torch modules then run, on the device.  ``norm`` (LayerNormnd in front of a
    single convolution): statistics from ``tcfd_row_moments_f64``, normalisation + affine folded into per-sample weights."""
    mods = [m for m in (lin1, lin2, skip_conv, norm) if m is not None]
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in [x, skip] + [p for m in mods for p in m.parameters()]):
        return None
    b, ci = x.shape[:2]
    co = lin2.out_channels
    cm = lin1.out_channels if lin1 is not None else ci
    if lin2.in_channels != cm or (lin1 is not None and lin1.in_channels != ci):
        return None
    P, T = x[0, 0].numel(), x.shape[-1]
    mode, s_t, sT = 0, None, 0
    if skip_conv is not None:
        if skip is None or skip.shape != x.shape or skip_conv.in_channels != ci or skip_conv.out_channels != co:
            return None
        mode, s_t = 1, skip.detach().contiguous()
    elif skip_last_slice:
        if skip is None or skip.shape[1] != co or skip.shape[2:-1] != x.shape[2:-1]:
            return None
        mode, s_t, sT = 2, skip.detach().contiguous(), skip.shape[-1]
    x = x.detach().contiguous()
    out = torch.empty(b, co, *x.shape[2:], dtype=torch.float64, device=x.device)
    mat = lambda conv, tr: (lambda w: (w.t() if tr else w).contiguous())(conv.weight.detach().reshape(conv.out_channels, conv.in_channels))
    w1 = mat(lin1, False) if lin1 is not None else None
    w2t = mat(lin2, True)
    wst = mat(skip_conv, True) if skip_conv is not None else None
    bias = lambda c: c.bias.detach().contiguous() if (c is not None and c.bias is not None) else None
    b1, b2, bs = bias(lin1), bias(lin2), bias(skip_conv)
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    w2_bs = b2_bs = 0
    if norm is not None:   # W'_b[c, o] = W[o, c] gamma_c rstd_b ,  b'_b[o] = bias[o] + sum_c W[o, c] (beta_c - gamma_c mu_b rstd_b)
        if lin1 is not None or norm.num_groups != 1 or norm.num_channels != ci:
            return None
        L = ci * P
        stats = torch.empty(b, 2, dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.tcfd_row_moments_f64(x.data_ptr(), stats.data_ptr(), b, L, stream), "tcfd_row_moments_f64")
        mu = stats[:, 0] / L
        rstd = torch.rsqrt((stats[:, 1] / L - mu * mu).clamp_min(0) + norm.eps)
        gamma = norm.weight.detach() if norm.weight is not None else torch.ones(ci, dtype=torch.float64, device=x.device)
        beta = norm.bias.detach() if norm.bias is not None else torch.zeros(ci, dtype=torch.float64, device=x.device)
        W = lin2.weight.detach().reshape(co, ci)
        w2t = (W.t()[None] * (gamma[None, :, None] * rstd[:, None, None])).contiguous()                 # (b, ci, co)
        fb = ((beta[None, :] - gamma[None, :] * (mu * rstd)[:, None])[:, None, :] * W[None]).sum(-1)
        b2 = (fb + lin2.bias.detach()[None] if lin2.bias is not None else fb).contiguous()              # (b, co)
        w2_bs, b2_bs = ci * co, co
    ptr = lambda t: t.data_ptr() if t is not None else None
    with torch.cuda.device(x.device):
        rc = lib.tcfd_fno_pointwise_f64(x.data_ptr(), ptr(s_t), out.data_ptr(), ptr(w1), ptr(b1), ptr(w2t), ptr(b2), ptr(wst),
                                        ptr(bs), b, ci, cm, co, P, T, sT, c1, c2, mode, w2_bs, b2_bs, stream)
    if rc == -1 and b"not instantiated" in lib.tcfd_last_error():
        _note_torch_modules(f"float64 pointwise block {ci} -> {cm} -> {co}")
        return None
    _lib.check(rc, "tcfd_fno_pointwise_f64")
    return out


def _lift_table_constants(qf: torch.Tensor):
    qd = qf.reshape(qf.shape[0] if qf.dim() == 2 else qf.shape[-4], -1).double()
    return qd.sum(dim=0).float().contiguous(), qd.sum(), (qd * qd).sum()


def _lift_fold(v1: torch.Tensor, q: torch.Tensor, norm: nn.GroupNorm, proj, consts=None):
    """First half of ``hip_lift_project``: the LayerNorm statistics of ``v1 + q`` from the one-channel input and the per-sample
    folded projection, ``proj(norm(v1 + q))[b, o] = sum_c w2t[b, c, o] (v1[b] + q[c]) + fb[b, o]`` (``tcfd_fno_lift_fold``, two
    launches).  Returns (v1 as (b, P), q as (C, P), w2t (b, C, co), fb (b, co), moments (b, 2)) or None when not covered."""
    if not v1.is_cuda or v1.dtype != torch.float32 or v1.shape[1] != 1 or not _is_pointwise(proj) or norm.num_groups != 1:
        return None
    if torch.is_grad_enabled() and (v1.requires_grad or q.requires_grad):
        return None
    for prm in list(norm.parameters()) + list(proj.parameters()):
        if prm.dtype != torch.float32 or prm.device != v1.device:
            raise TypeError(f"fp32 input on {v1.device} but a {prm.dtype} parameter on {prm.device}")
    q = q.reshape(-1, *q.shape[-3:]) if q.dim() == 5 else q
    C = q.shape[0]
    if q.shape[1:] != v1.shape[2:] or norm.num_channels != C or proj.in_channels != C:
        return None
    b, co = v1.shape[0], proj.out_channels
    P = v1[0, 0].numel()
    dev = v1.device
    qf = q.detach().to(device=dev, dtype=torch.float32).reshape(C, P).contiguous()
    vf = v1.detach().reshape(b, P).contiguous()
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    if consts is None:   # constants of the table: channel sum per point, total sum, total sum of squares
        consts = _lift_table_constants(qf)
    qs, sq, sq2 = consts
    # the three per-sample sums over v and the per-sample folded weights in two launches (tcfd_fno_lift_fold; this was a GEMV
    # and ~25 small tensor-op launches per forward: 0.1 ms of a 5 ms forward at config 5)
    w2t = torch.empty(b, C, co, dtype=torch.float32, device=dev)
    fb = torch.empty(b, co, dtype=torch.float32, device=dev)
    moments = torch.empty(b, 2, dtype=torch.float64, device=dev)
    scratch = torch.empty(b, 3, dtype=torch.float64, device=dev)
    ptr = lambda t: t.detach().contiguous().data_ptr() if t is not None else None
    Wc = proj.weight.detach().reshape(co, C).contiguous()
    with torch.cuda.device(dev):
        _lib.check(lib.tcfd_fno_lift_fold(vf.data_ptr(), qs.data_ptr(), sq.data_ptr(), sq2.data_ptr(), Wc.data_ptr(), ptr(proj.bias),
                                          ptr(norm.weight), ptr(norm.bias), float(norm.eps), w2t.data_ptr(), fb.data_ptr(),
                                          moments.data_ptr(), scratch.data_ptr(), b, C, co, P, stream), "tcfd_fno_lift_fold")
    return vf, qf, w2t, fb, moments


def hip_lift_project(v1: torch.Tensor, q: torch.Tensor, norm: nn.GroupNorm, proj, consts=None) -> Optional[torch.Tensor]:
    """``proj(norm(v1 + q))`` of the lifting operator (fno/sfno.py:252-254) without ever forming ``v1 + q``.

    v1 (b, 1, X, Y, T) is the single input channel, q (1, C, X, Y, T) the positional-encoding table that the reference
    adds to it by broadcasting.  The LayerNorm statistics of the (C, X, Y, T) block of a sample follow from three
    reductions over the ONE-channel input and constants of the table,
        sum = C sum(v) + sum(q),   sum of squares = C sum(v^2) + 2 sum_p v_p (sum_c q_cp) + sum(q^2),
    and the projection kernel rebuilds v + q[c] in registers (``pe`` mode of ``tcfd_fno_pointwise``): 84 MB + a 26 MB
    L2-resident table are read instead of writing and re-reading an 839 MB tensor twice (config 5).  ``consts`` are the
    table's constants when the caller has them cached (``SpaceTimePositionalEncoding.table_constants``).  Training (the
    parameters of ``norm`` / ``proj`` require grad, the input is data): the same forward behind ``_LiftProjectFn``, whose
    backward forms ``v1 + q`` once and reuses the forward's moments.  Returns None when the combination is not covered
    (an input that itself requires grad included)."""
    training = torch.is_grad_enabled() and any(p.requires_grad for m in (norm, proj) for p in m.parameters())
    if training and os.environ.get("TCFD_LIFT_PROJECT_GRAD", "1") == "0":
        return None
    fold = _lift_fold(v1, q, norm, proj, consts)
    if fold is None:
        return None
    vf, qf, w2t, fb, moments = fold
    b, co, C = v1.shape[0], proj.out_channels, qf.shape[0]
    P = v1[0, 0].numel()
    dev = v1.device
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = torch.empty(b, co, *v1.shape[2:], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.tcfd_fno_pointwise(vf.data_ptr(), None, out.data_ptr(), None, None, w2t.data_ptr(), fb.data_ptr(), None, None,
                                    b, C, C, co, P, v1.shape[-1], 0, 0, 0, 0, C * co, co, qf.data_ptr(), stream)
    if rc == -1 and b"not instantiated" in lib.tcfd_last_error():
        return None
    _lib.check(rc, "tcfd_fno_pointwise")
    if training:
        dims = (ctypes.c_int * 6)()      # layout query of the per-sample weight-gradient pass (no data)
        if lib.tcfd_fno_pointwise_bwd(None, None, None, None, None, None, None, None, None, None, None, None, 0, dims, b, C, C, co,
                                      P, 0, 0, 0, 0, 0, 1, None) != 0:
            return None
        # (moments: sum and sum of squares of the (C, X, Y, T) block of every sample, from the fold kernel)
        return _LiftProjectFn.apply(out, vf, qf, moments, float(norm.eps), tuple(v1.shape[2:]), proj.weight, proj.bias,
                                    norm.weight, norm.bias)
    return out


class _LiftProjectFn(torch.autograd.Function):
    """``proj(norm(v1 + q))`` under autograd with the forward of ``hip_lift_project``: v1 + q is never formed, neither in the
    forward nor in the backward (the weight-gradient kernel rebuilds it in registers, ``tcfd_fno_pointwise_bwd_pe``); the
    LayerNorm moments come from the forward and the input gradient is skipped (v1 and the table are data)."""

    @staticmethod
    def forward(ctx, out, vf, qf, moments, eps, mesh, w, bias, gamma, beta):
        ctx.eps, ctx.mesh = eps, mesh
        ctx.present = [t is not None for t in (w, bias, gamma, beta)]
        ctx.save_for_backward(vf, qf, moments, *[t for t in (w, bias, gamma, beta) if t is not None])
        return out.view_as(out)

    @staticmethod
    def backward(ctx, dout):
        vf, qf, moments, *rest = ctx.saved_tensors
        it = iter(rest)
        w, bias, gamma, beta = [next(it) if p else None for p in ctx.present]
        res = _hip_norm_proj_backward(ctx.eps, dout, vf, w, bias, gamma, beta, need_dx=False, moments=moments, table=qf)
        if res is None:
            raise _lib.TcfdError("LayerNorm-projection backward kernel not available for a block admitted to the fused path")
        need = ctx.needs_input_grad[6:]
        g_w, g_b, g_gamma, g_beta = res[4], res[5], res[8], res[9]
        return (None, None, None, None, None, None, *[g if n else None for g, n in zip((g_w, g_b, g_gamma, g_beta), need)])


# ----------------------------------------------------------------------------- spectral convolutions
class SpectralConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, modes: List[int], dim: int, bias: bool = False,
                 norm: str = "backward") -> None:
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.dim = dim
        self.bias = bias
        self.norm = norm
        assert len(modes) == dim, "modes should match the dimension"
        size = [in_channels, out_channels, *modes, 2]
        gain = 0.5 / (in_channels * out_channels)
        self._initialize_weights(size, gain)
        self._set_complex_matmul_nd(dim)

    def _initialize_weights(self, size, gain=1e-4):
        n_blocks = 2 * (self.dim - 1)
        self.weight = nn.ParameterList([nn.Parameter(gain * torch.rand(*size)) for _ in range(n_blocks)])
        if self.bias:
            self.bias = nn.ParameterList([nn.Parameter(gain * torch.zeros(*size[2:])) for _ in range(n_blocks)])

    def _bias_list(self):
        return list(self.bias) if isinstance(self.bias, nn.ParameterList) else None

    # -- the dimension-generic template of fno/base.py:114-237: 2 (dim - 1) weight blocks, a channel contraction over any
    #    number of mesh axes, forward = rfftn -> spectral_conv (the subclass's) -> irfftn.  The (2+1)-D subclasses below
    #    replace ``forward`` by the fused HIP kernels; any other dimension runs the two transforms as dense matrix products
    #    on the device (dense_fft.py) around the subclass's ``spectral_conv``.
    @staticmethod
    def complex_matmul(x, w, **kwargs):
        """(b, c_i, *mesh), (c_i, c_o, *mesh) -> (b, c_o, *mesh)"""
        return torch.einsum("bi...,io...->bo...", x, w)

    def _set_complex_matmul_nd(self, dim: int = None):
        """Bind ``complex_matmul`` to the einsum of exactly ``dim`` mesh axes ("bixy,ioxy->boxy" for dim = 2)."""
        dim = self.dim if dim is None else dim
        assert dim >= 1
        mesh = "".join(chr(ord("z") - k) for k in range(dim - 1, -1, -1))
        equation = f"bi{mesh},io{mesh}->bo{mesh}"
        self.complex_matmul = functools.partial(torch.einsum, equation)      # picklable, unlike a closure

    def spectral_conv(self, vhat, *fft_mesh_size, **kwargs):
        raise NotImplementedError("Subclasses must implement spectral_conv() to perform spectral convolution")

    def forward(self, v, out_mesh_size=None, **kwargs):
        from .dense_fft import irfftn_dense, rfftn_dense

        if not v.is_cuda:
            raise _lib.TcfdError("expected a HIP device tensor (torch-cfd_amd has no CPU fallback)")
        mesh_size = list(v.shape[2:])
        if len(mesh_size) != self.dim:
            raise ValueError(f"expected (b, C) + {self.dim} mesh axes, got {tuple(v.shape)}")
        out_mesh_size = mesh_size if out_mesh_size is None else [int(n) for n in out_mesh_size]
        fft_mesh_size = mesh_size.copy()
        fft_mesh_size[-1] = mesh_size[-1] // 2 + 1
        v_hat = rfftn_dense(v, self.dim, self.norm)
        v_hat = self.spectral_conv(v_hat, *fft_mesh_size, **kwargs)
        return irfftn_dense(v_hat, out_mesh_size, self.norm)


class SpectralConvS(SpectralConv):
    def __init__(self, in_channels: int, out_channels: int, modes_x: int, modes_y: int, modes_t: int, dim: int = 3,
                 bias: bool = False, delta: float = 1, norm="backward") -> None:
        super().__init__(in_channels=in_channels, out_channels=out_channels, modes=(modes_x, modes_y, modes_t),
                         dim=dim, bias=bias, norm=norm)
        self.modes_x, self.modes_y, self.modes_t = modes_x, modes_y, modes_t
        self.delta = delta

    @property
    def modes(self):
        return (self.modes_x, self.modes_y, self.modes_t)

    def _plain_args(self, v, out_steps=None):
        """(weights, bias, delta, modes, t_pad, t_out, t_keep, norm) of ``forward(v)`` -- what ``hip_spectral_layer`` needs to
        run the convolution piecewise -- or None when this call is not the plain transform / contract / inverse chain."""
        T = v.shape[-1]
        return list(self.weight), self._bias_list(), self.delta, self.modes, 0, T, T, self.norm

    def spectral_conv(self, vh, kx: int = None, ky: int = None, kt: int = None):
        """Contraction on an ALREADY TRUNCATED spectrum (b, Ci, 2mx, 2my, mt) -> (b, Co, 2mx, 2my, mt)."""
        return hip_contract(vh, list(self.weight), self._bias_list(), self.delta, self.modes)

    def forward(self, v, out_mesh_size=None, **kwargs):
        t_out = None
        if out_mesh_size is not None:
            t_out = out_mesh_size[-1]
            if tuple(out_mesh_size[:2]) != tuple(v.shape[-3:-1]):
                return self._resampled(v, tuple(int(n) for n in out_mesh_size))
        return hip_spectral_conv(v, list(self.weight), self._bias_list(), self.delta, self.modes, t_out=t_out,
                                 norm=self.norm)

    def _resampled(self, v, out_size):
        """``irfftn(spectrum, s=out_size)`` with a spatial size other than the input's (fno/base.py:229-237): torch pads /
        trims the spectrum array at its END, so the high-frequency block stays at the array indices it had on the input
        grid; the inverse plan reproduces exactly that placement (``tcfd_fno_plan_create_resample``)."""
        b, c, X, Y, T = v.shape
        Xo, Yo, To = out_size
        mx, my, mt = self.modes
        wants_grad = torch.is_grad_enabled() and (v.requires_grad or any(p.requires_grad for p in self.parameters()))
        ok = not wants_grad and _library_takes(X, Y, T, 0, T, self.modes, T, v.device, v.dtype) and _fused_xy(Xo, Yo)
        if ok:      # ... and the inverse plan on the output grid
            inv = _plan((Xo, Yo, T, 0, To, mx, my, mt, X, Y), v.device, _real_of(v.dtype))
            ok = bool(inv.lib.tcfd_fno_plan_supports(inv.handle, int(To)))
        if not ok:
            # gradients of a resampled layer (super-resolution fine-tuning differentiates it, fno/base.py:229-237) and grids
            # off the fused kernels: the dense pruned transforms (differentiable)
            return dense_spectral_conv(v, list(self.weight), self._bias_list(), self.delta, self.modes, 0, To, To, self.norm,
                                       out_xy=(Xo, Yo))
        vh, _ = hip_truncated_rfftn(v, self.modes, norm=self.norm)
        oh = hip_contract(vh, list(self.weight), self._bias_list(), self.delta, self.modes)
        _, scale = _norm_scales(self.norm, X * Y * T, Xo * Yo * To)
        return hip_truncated_irfftn(oh, inv, To, scale=scale)


class SpectralConvT(SpectralConvS):
    def __init__(self, in_channels: int, out_channels: int, modes_x: int, modes_y: int, modes_t: int,
                 delta: float = 1e-1, out_steps: int = None, norm: str = "backward", bias: bool = True,
                 temporal_padding: bool = False, postprocess: nn.Module = nn.Identity(), **kwargs) -> None:
        super().__init__(in_channels, out_channels, modes_x, modes_y, modes_t, norm=norm, delta=delta, bias=bias)
        self.out_steps = out_steps
        self.temporal_padding = temporal_padding
        self.postprocess = postprocess

    def _plain_args(self, v, out_steps=None):
        if out_steps is None and self.out_steps is not None:
            out_steps = self.out_steps
        if out_steps is None or not isinstance(self.postprocess, nn.Identity):
            return None
        t_pad = v.size(-1) if self.temporal_padding else 0
        return list(self.weight), self._bias_list(), self.delta, self.modes, t_pad, out_steps + t_pad, out_steps, self.norm

    def forward(self, v, out_steps: int = None, keep_steps: int = None):
        """``keep_steps`` (an extension; the reference's signature ends at ``out_steps``): return only the LAST ``keep_steps`` of
        the ``out_steps`` steps -- the inverse transform then never produces the others (the output operator drops the first
        one, fno/sfno.py:326: a strided slice + copy of the whole output otherwise)."""
        if out_steps is None and self.out_steps is not None:
            out_steps = self.out_steps
        t_pad = v.size(-1) if self.temporal_padding else 0
        if out_steps is None:      # (the reference fails later, on `out_steps + t_pad`; say what is missing)
            raise ValueError("SpectralConvT needs out_steps: pass it to forward() or to the constructor")
        if keep_steps is not None:
            if not 0 < keep_steps <= out_steps:
                raise ValueError(f"keep_steps = {keep_steps} outside (0, {out_steps}]")
            if isinstance(self.postprocess, nn.Identity):
                return hip_spectral_conv(v, list(self.weight), self._bias_list(), self.delta, self.modes, t_pad=t_pad,
                                         t_out=out_steps + t_pad, t_keep=keep_steps, norm=self.norm)
            return self.forward(v, out_steps)[..., -keep_steps:]
        if not isinstance(self.postprocess, nn.Identity):
            # spectrum post-processing (Helmholtz projection for out_dim = 2): the projection is diagonal in k,
            # so it acts on the kept modes only -- transform, contract, project, inverse-transform
            if not v.is_cuda or v.dtype not in (torch.float32, torch.float64):
                raise _lib.TcfdError("expected an fp32 / fp64 HIP device tensor (torch-cfd_amd has no CPU fallback)")
            if not _library_takes(v.shape[2], v.shape[3], v.shape[4], t_pad, out_steps + t_pad, self.modes, out_steps, v.device,
                                  v.dtype):
                post = (lambda oh: self.postprocess.forward_truncated(oh, self.modes, v.shape[-3])) if hasattr(
                    self.postprocess, "forward_truncated") else self.postprocess
                return dense_spectral_conv(v, list(self.weight), self._bias_list(), self.delta, self.modes, t_pad,
                                           out_steps + t_pad, out_steps, self.norm, post=post)
            if torch.is_grad_enabled() and (v.requires_grad or any(p.requires_grad for p in self.parameters())):
                post = (lambda oh: self.postprocess.forward_truncated(oh, self.modes, v.shape[-3])) if hasattr(
                    self.postprocess, "forward_truncated") else self.postprocess
                return hip_spectral_conv_autograd(v, list(self.weight), self._bias_list(), self.delta, self.modes, t_pad,
                                                  out_steps + t_pad, out_steps, self.norm, post=post)
            vh, plan = hip_truncated_rfftn(v, self.modes, t_pad=t_pad, t_out=out_steps + t_pad, norm=self.norm)
            oh = hip_contract(vh, list(self.weight), self._bias_list(), self.delta, self.modes)
            oh = self.postprocess.forward_truncated(oh, self.modes, v.shape[-3]) if hasattr(
                self.postprocess, "forward_truncated") else self.postprocess(oh)
            return hip_truncated_irfftn(oh, plan, out_steps, norm=self.norm)
        return hip_spectral_conv(v, list(self.weight), self._bias_list(), self.delta, self.modes, t_pad=t_pad,
                                 t_out=out_steps + t_pad, t_keep=out_steps, norm=self.norm)


class SpectralConv3d(nn.Module):
    """The original FNO3d Fourier layer: 4 complex weight blocks ``weights1..4``."""

    def __init__(self, in_channels, out_channels, modes1, modes2, modes3):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.modes1, self.modes2, self.modes3 = modes1, modes2, modes3
        self.scale = 1 / (in_channels * out_channels)
        shape = (in_channels, out_channels, modes1, modes2, modes3)
        for k in (1, 2, 3, 4):
            setattr(self, f"weights{k}", nn.Parameter(self.scale * torch.rand(*shape, dtype=torch.cfloat)))

    def forward(self, x):
        w = [self.weights1, self.weights2, self.weights3, self.weights4]
        return hip_spectral_conv(x, w, None, 1.0, (self.modes1, self.modes2, self.modes3))


# ----------------------------------------------------------------------------- SFNO
class SpaceTimePositionalEncoding(nn.Module):
    """Channels [x, y, t, e^{beta t} sin/cos(pi (k+1) t) ...] added to the single input channel
    (fno/sfno.py:25-113; the random-feature variant :62-88 is supported too)."""

    def __init__(self, modes_x: int = 16, modes_y: int = 16, modes_t: int = 5, num_channels: int = 20,
                 input_shape=(64, 64, 10), spatial_random_feats: bool = False, max_time_steps: int = 100,
                 time_exponential_scale: float = 1e-2, **kwargs):
        super().__init__()
        assert num_channels % 2 == 0 and num_channels > 3
        self.num_channels = num_channels
        self.max_time_steps = max_time_steps
        self.time_exponential_scale = time_exponential_scale
        self.modes_x, self.modes_y, self.modes_t = modes_x, modes_y, modes_t
        self.spatial_random_feats = spatial_random_feats
        self._build(*input_shape)
        if spatial_random_feats:
            self.proj = nn.Conv3d(modes_x * modes_y * modes_t + 3, num_channels, kernel_size=1)
        else:
            self.proj = nn.Identity()

    @staticmethod
    def _wave(order: torch.Tensor, coord: torch.Tensor, sine_when_even: torch.Tensor) -> torch.Tensor:
        """Rows sin(pi order_r coord) / cos(pi order_r coord), sine where ``sine_when_even`` holds: (R, len(coord))."""
        phase = (torch.pi * order)[:, None] * coord[None, :]
        return torch.where(sine_when_even[:, None], torch.sin(phase), torch.cos(phase))

    def _build(self, nx, ny, nt):
        """The (1, C, nx, ny, nt) table: coordinates x, y in [0, 1], t = (1..nt) / max_time_steps, then either
        C - 3 purely temporal channels e^{beta t} {sin, cos}(pi (k+1) t), k = 0..C-4 (sine for even k), broadcast
        over space, or -- random-feature variant -- the mx my mt separable products
        e^{beta t} b_i(x) b_j(y) b_k(t) / (i j k), b_r = sine for even r, cosine for odd r, frequency pi r."""
        x = torch.linspace(0, 1, nx)
        y = torch.linspace(0, 1, ny)
        t = torch.linspace(0, 1, self.max_time_steps + 1)[1: nt + 1]
        growth = torch.exp(self.time_exponential_scale * t)
        coords = torch.stack(torch.meshgrid(x, y, t, indexing="ij"))
        if self.spatial_random_feats:
            def basis(count, coord):
                r = torch.arange(1, count + 1)
                return self._wave(r.to(coord.dtype), coord, r % 2 == 0) / r[:, None]

            bx, by, bt = basis(self.modes_x, x), basis(self.modes_y, y), basis(self.modes_t, t) * growth
            feats = torch.einsum("ix,jy,kt->ijkxyt", bx, by, bt).reshape(-1, nx, ny, nt)
        else:
            k = torch.arange(self.num_channels - 3)
            feats = (self._wave((k + 1).to(t.dtype), t, k % 2 == 0) * growth)[:, None, None, :].expand(-1, nx, ny, -1)
        self.pe = torch.cat([coords, feats]).unsqueeze(0).contiguous()

    def encoding(self, v):
        """The (1, C, X, Y, T) table that ``forward`` adds to v."""
        if self.pe is None or self.pe.shape[-3:] != v.shape[-3:]:
            self._build(*v.shape[-3:])
            self._consts = None
        if self.pe.device != v.device or self.pe.dtype != v.dtype:
            self.pe = self.pe.to(device=v.device, dtype=v.dtype)  # keep the table resident on the device
            self._consts = None
        return self.proj(self.pe)

    def table_constants(self, v):
        """(channel sum per point, total sum, total sum of squares) of the table, cached with it; None when the table
        goes through a learned projection (then it changes with the weights and is reduced per call)."""
        if not isinstance(self.proj, nn.Identity):
            return None
        self.encoding(v)
        if getattr(self, "_consts", None) is None:
            self._consts = _lift_table_constants(self.pe.to(torch.float32))
        return self._consts

    def forward(self, v):
        return v + self.encoding(v)


class HelmholtzProjection(nn.Module):
    """Divergence-free projection in Fourier space, w^ = u^ - grad div u^ / lap (fno/sfno.py:116-193);
    element-wise device ops."""

    def __init__(self, n_grid: int = 64, diam: float = 2 * torch.pi, dtype: torch.dtype = torch.float32):
        super().__init__()
        self.n_grid, self.diam = n_grid, diam
        self._update_fft_mesh(n_grid, diam, dtype)

    def _update_fft_mesh(self, n, diam=None, dtype=torch.float32):
        diam = diam if diam is not None else self.diam
        k = torch.fft.fftfreq(n, d=diam / n)
        kx, ky = torch.meshgrid([k, k], indexing="ij")
        lap = -4 * (torch.pi**2) * (abs(kx) ** 2 + abs(ky) ** 2)
        lap[..., 0, 0] = 1
        dev = self.lap.device if hasattr(self, "lap") else None
        for name, val in (("lap", lap), ("kx", kx), ("ky", ky)):
            val = val.to(dtype).clone()  # meshgrid returns expanded views; buffers must own their memory
            val = val.to(dev) if dev is not None else val
            if hasattr(self, name):
                setattr(self, name, val)
            else:
                self.register_buffer(name, val)

    @staticmethod
    def div(uhat, fft_mesh):
        kx, ky = (z[None, :, :, None] for z in fft_mesh)
        return 2j * torch.pi * (uhat[:, 0] * kx + uhat[:, 1] * ky)

    @staticmethod
    def grad(uhat, fft_mesh):
        kx, ky = (z[None, :, :, None] for z in fft_mesh)
        return torch.stack((2j * torch.pi * kx * uhat, 2j * torch.pi * ky * uhat), dim=1)

    def forward(self, uhat):
        nx = uhat.shape[2]
        if nx != self.n_grid:
            self._update_fft_mesh(nx)
            self.n_grid = nx
        mesh = (self.kx, self.ky)
        g = self.grad(self.div(uhat, mesh), mesh)
        return uhat - g / self.lap[None, None, :, :, None]

    def forward_truncated(self, uhat, modes, n_grid: int):
        """The same projection on a spectrum stored only at the kept modes (b, 2, 2mx, 2my, mt): the projection
        is diagonal in k, so the wavenumber tables are simply restricted to rows [:mx] + [-mx:], cols [:my] + [-my:]."""
        if n_grid != self.n_grid:
            self._update_fft_mesh(n_grid)
            self.n_grid = n_grid
        mx, my, _ = modes
        rows = torch.cat([torch.arange(mx), torch.arange(n_grid - mx, n_grid)]).to(uhat.device)
        cols = torch.cat([torch.arange(my), torch.arange(n_grid - my, n_grid)]).to(uhat.device)
        sub = lambda z: z.to(uhat.device)[rows][:, cols]
        kx, ky, lap = sub(self.kx), sub(self.ky), sub(self.lap)
        g = self.grad(self.div(uhat, (kx, ky)), (kx, ky))
        return uhat - g / lap[None, None, :, :, None]


_LIFT_TABLE_MODES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()   # LiftingOperator -> (key, table modes, last table slice)


class LiftingOperator(nn.Module):
    def __init__(self, width: int, modes_x: int, modes_y: int, modes_t: int, latent_steps: int = 10,
                 norm: str = "backward", activation: ActivationType = "GELU", beta: float = 0.1,
                 spatial_random_feats: bool = False, channel_expansion: int = 4, nonlinear: bool = True, **kwargs):
        super().__init__()
        pe_modes_t = modes_t - 1 if modes_t % 2 != 0 else modes_t
        self.pe = SpaceTimePositionalEncoding(modes_x // 2, modes_y // 2, pe_modes_t // 2, num_channels=width,
                                              time_exponential_scale=beta,
                                              spatial_random_feats=spatial_random_feats)
        in_channels = self.pe.num_channels
        self.norm = LayerNormnd(in_channels)
        self.proj = nn.Conv3d(in_channels, width, kernel_size=1)
        self.sconv = SpectralConvT(width, width, modes_x, modes_y, modes_t, out_steps=latent_steps, norm=norm,
                                   bias=False)
        self.latent_steps = latent_steps
        if nonlinear:
            self.activation = getattr(nn, activation)()
            self.mlp = PointwiseFFN(width, width, channel_expansion * width, activation)
        else:
            self.activation = nn.Identity()
            self.mlp = nn.Conv3d(width, width, kernel_size=1)

    def _through_the_spectrum(self, vin):
        """Inference form of the whole operator that never forms the projected tensor v0 = proj(norm(pe(v))) (b, width, X, Y, T):
        its kept modes are an affine map of the kept modes of the ONE input channel (``tcfd_fno_lift_spectrum``: the transform
        is linear, the projection a per-sample affine map of v + table), and the tail needs only v0's last time slice
        (fno/sfno.py:258-259).  At config 5: one 1-channel transform + a 30 MB combine instead of the projection pass (0.18 ms,
        writes 839 MB) and a 10-channel transform (0.19 ms, reads them back).  Same result to fp32 rounding.  None when not
        covered (gradients, dtype, a grid off the FFT kernels, a convolution subclass)."""
        conv = self.sconv
        if (os.environ.get("TCFD_LIFT_SPECTRUM", "1") == "0" or not vin.is_cuda or vin.dtype != torch.float32 or vin.dim() != 5
                or vin.shape[1] != 1 or vin.shape[0] == 0 or not _fused_xy(vin.shape[2], vin.shape[3])):
            return None
        if torch.is_grad_enabled() and (vin.requires_grad or any(p.requires_grad for p in self.parameters())):
            return None
        cls = type(conv)
        if cls.forward is not SpectralConvT.forward or cls.spectral_conv is not SpectralConvS.spectral_conv \
                or cls._plain_args is not SpectralConvT._plain_args or any(p.dtype != torch.float32 for p in conv.parameters()):
            return None
        cargs = conv._plain_args(vin, None)
        if cargs is None:
            return None
        weights, bias, delta, modes, t_pad, t_out, t_keep, norm = cargs
        b, _, X, Y, T = vin.shape
        if not _library_takes(X, Y, T, t_pad, t_out, modes, t_keep, vin.device):
            return None
        q = self.pe.encoding(vin)
        fold = _lift_fold(vin, q, self.norm, self.proj, consts=self.pe.table_constants(vin))
        if fold is None:
            return None
        vf, qf, w2t, fb, _ = fold
        C, co = qf.shape[0], self.proj.out_channels
        if C > 32 or co != conv.in_channels:
            return None
        dev = vin.device
        # kept modes of the table channels and of the constant field, and the table's last time slice: input independent
        # (a table behind a learned projection -- spatial_random_feats -- changes with its weights: formed per call then)
        fixed_table = isinstance(self.pe.proj, nn.Identity)
        key = (dev, X, Y, T, tuple(modes), t_pad, t_out, norm, qf.data_ptr(), q._version)
        # (kept OUTSIDE the module, keyed weakly by it: device tensors in a plain attribute would ride along with copy.deepcopy and
        #  torch.save(model) and stay behind on .to())
        cached = _LIFT_TABLE_MODES.get(self) if fixed_table else None
        if cached is None or cached[0] != key:
            with torch.no_grad():
                fields = torch.cat([qf.view(C, 1, X, Y, T), torch.ones(1, 1, X, Y, T, dtype=torch.float32, device=dev)])
                th, _ = hip_truncated_rfftn(fields, modes, t_pad=t_pad, t_out=t_out, norm=norm)
                q_last = qf.view(C, X * Y, T)[..., -1].contiguous()
            cached = (key, th.reshape(C + 1, -1).contiguous(), q_last)
            # (not kept when formed inside a stream capture: its memory then belongs to the graph's private pool)
            if fixed_table and not torch.cuda.is_current_stream_capturing():
                _LIFT_TABLE_MODES[self] = cached
            else:
                _LIFT_TABLE_MODES.pop(self, None)
        _, table, q_last = cached
        vh, plan = hip_truncated_rfftn(vin, modes, t_pad=t_pad, t_out=t_out, norm=norm)
        K = vh[0, 0].numel()
        v0h = torch.empty(b, co, *vh.shape[2:], dtype=torch.complex64, device=dev)
        lib = _lib.load()
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        v_last = vf.view(b, X * Y, T)[..., -1].contiguous()
        skip = torch.empty(b, co, X, Y, 1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.tcfd_fno_lift_spectrum(vh.data_ptr(), table.data_ptr(), w2t.data_ptr(), fb.data_ptr(), v0h.data_ptr(),
                                                  b, C, co, K, stream), "tcfd_fno_lift_spectrum")
            # v0[..., -1:] alone: the projection kernel on the last time slice (a tenth of its points)
            rc = lib.tcfd_fno_pointwise(v_last.data_ptr(), None, skip.data_ptr(), None, None, w2t.data_ptr(), fb.data_ptr(), None,
                                        None, b, C, C, co, X * Y, 1, 0, 0, 0, 0, C * co, co, q_last.data_ptr(), stream)
        if rc == -1 and b"not instantiated" in lib.tcfd_last_error():
            return None
        _lib.check(rc, "tcfd_fno_pointwise")
        oh = hip_contract(v0h, weights, bias, delta, modes)
        lin = (self.mlp.linear1, self.mlp.activation, self.mlp.linear2) if isinstance(self.mlp, PointwiseFFN) else (None, None, self.mlp)
        x1 = hip_truncated_irfftn(oh, plan, t_keep, norm=norm)
        if isinstance(self.mlp, PointwiseFFN):
            out = hip_pointwise(x1, self.mlp.linear1, self.mlp.activation, self.mlp.linear2, skip=skip, act2=self.activation,
                                skip_last_slice=True)
        else:
            out = hip_pointwise(x1, None, None, self.mlp, skip=skip, act2=self.activation, skip_last_slice=True)
        return out if out is not None else self.activation(skip + self.mlp(x1))

    def forward(self, v):
        assert self.latent_steps <= v.size(-1)
        vin = v
        out = self._through_the_spectrum(vin)
        if out is not None:
            return out
        v = hip_lift_project(vin, self.pe.encoding(vin), self.norm, self.proj,
                             consts=self.pe.table_constants(vin)) if vin.shape[1] == 1 else None
        if v is None:
            vp = self.pe(vin)
            v = hip_pointwise(vp, None, None, self.proj, norm=self.norm)  # LayerNormnd folded into the projection
            if v is None:
                v = self.proj(self.norm(vp))
        lin = (self.mlp.linear1, self.mlp.activation, self.mlp.linear2) if isinstance(self.mlp, PointwiseFFN) else (None, None, self.mlp)
        out = hip_spectral_layer(self.sconv, v, *lin, act2=self.activation, skip_last_slice=True)
        if out is not None:             # training: convolution + tail as one autograd node
            return out
        x1 = self.sconv(v)
        if isinstance(self.mlp, PointwiseFFN):
            out = hip_pointwise(x1, self.mlp.linear1, self.mlp.activation, self.mlp.linear2, skip=v,
                                act2=self.activation, skip_last_slice=True)
        else:
            out = hip_pointwise(x1, None, None, self.mlp, skip=v, act2=self.activation, skip_last_slice=True)
        if out is not None:
            return out
        return self.activation(v[..., -1:] + self.mlp(x1))


class _OutHeadFn(torch.autograd.Function):
    """Channel reduction + output operator of the SFNO (fno/sfno.py:313-328, 618-620) as ONE autograd node on the inference
    kernels: the forward values come from ``OutConv.fused_forward`` (``tcfd_fno_reduce_frames``: the reduction writes its
    latent steps behind the last input frame, no ``torch.cat``; ``tcfd_fno_inverse_trunc_residual``: the residual frame is
    added in the store loop, no slice copy, no ``add``), the backward chains the adjoints the separate nodes use --
    ``_inv_trunc_vjp``, ``_contract_vjp``, ``_fwd_trunc_vjp`` and the single-layer pointwise backward for the reduction.
    The input frames ``v_res`` get no gradient here (the caller falls back to the composed path when they need one)."""

    @staticmethod
    def forward(ctx, out, vh, cfg, v, red_w, red_b, *conv_params):
        ctx.cfg = cfg
        ctx.has_rb = red_b is not None
        ctx.save_for_backward(vh, v, red_w, *([red_b] if red_b is not None else []), *conv_params)
        return out.view_as(out)

    @staticmethod
    def backward(ctx, dout):
        saved = list(ctx.saved_tensors)
        vh, v, red_w = saved[:3]
        red_b = saved[3] if ctx.has_rb else None
        conv_params = saved[4 if ctx.has_rb else 3:]
        fwd_cfg, contract_cfg, inv_cfg = ctx.cfg
        need = ctx.needs_input_grad
        g_oh = _inv_trunc_vjp(dout.unsqueeze(1), inv_cfg)
        gv_h, grads = _contract_vjp(g_oh, vh, conv_params, contract_cfg, True, need[6:])
        g_frames = _fwd_trunc_vjp(gv_h, fwd_cfg)                      # (b, 1, X, Y, T + 1): frame 0 is the input frame's
        g_red = g_frames[..., 1:].contiguous()
        spec = (False, None, None, 0, None)
        hip = _hip_pointwise_backward(spec, g_red, v, None, None, None, red_w, red_b, None, None, None, None, need_dx=bool(need[3]))
        if hip is None:       # (a width the pointwise backward is not instantiated for: the same sums as tensor ops)
            b, C = v.shape[:2]
            g2 = g_red.reshape(b, 1, -1)
            dv = (red_w.detach().reshape(1, C, 1) * g2).view_as(v) if need[3] else None
            gw = torch.einsum("bop,bcp->oc", g2, v.detach().reshape(b, C, -1)).reshape(red_w.shape)
            gb = g2.sum(dim=(0, 2)) if red_b is not None else None
        else:
            dv, gw, gb = hip[0], hip[4], hip[5]
        return (None, None, None, dv if need[3] else None, gw if need[4] else None, gb if (red_b is not None and need[5]) else None,
                *grads)


class OutConv(nn.Module):
    def __init__(self, modes_x: int, modes_y: int, modes_t: int, delta: float = 0.1, out_dim: int = 1,
                 diam: float = 1, n_grid: int = 64, out_steps: int = None, spatial_padding: int = 0,
                 temporal_padding: bool = True, norm: str = "backward", **kwargs):
        super().__init__()
        self.size = [out_dim, out_dim, modes_x, modes_y, modes_t]
        postprocess = HelmholtzProjection(n_grid=n_grid, diam=diam) if out_dim == 2 else nn.Identity()
        self.conv = SpectralConvT(*self.size, norm=norm, delta=delta, out_steps=out_steps, bias=True,
                                  temporal_padding=temporal_padding, postprocess=postprocess)
        self.n_grid, self.norm, self.delta = n_grid, norm, delta
        self.spatial_padding, self.temporal_padding = spatial_padding, temporal_padding

    def fused_forward(self, v, v_res, reduction, out_steps: int):
        """``self(reduction(v), v_res, out_steps)`` for the forward-only fp32 single-output-channel case without the glue
        passes of fno/sfno.py:313-328: the channel reduction writes its latent steps behind the last input frame
        (``tcfd_fno_reduce_frames``: no ``torch.cat``), the inverse transform produces only the kept steps and adds the residual
        frame in its store loop (``tcfd_fno_inverse_trunc_residual``: no slice copy, no ``add``).  None when not covered."""
        conv = self.conv
        training = (torch.is_grad_enabled() and (v.requires_grad or v_res.requires_grad or any(p.requires_grad for p in self.parameters())
                                                 or any(p.requires_grad for p in reduction.parameters())))
        if training and (v_res.requires_grad or os.environ.get("TCFD_FNO_FUSED_OUT_TRAIN", "1") == "0"):
            return None               # a gradient for the input frames themselves: the composed path
        if (not v.is_cuda or v.dtype != torch.float32 or v.dim() != 5 or v_res.dim() != 4 or v_res.dtype != torch.float32
                or self.spatial_padding > 0 or self.size[0] != 1 or not _is_pointwise(reduction) or reduction.out_channels != 1
                or type(conv).forward is not SpectralConvT.forward or not isinstance(conv.postprocess, nn.Identity)
                or os.environ.get("TCFD_FNO_FUSED_OUT", "1") == "0"):
            return None
        b, C, X, Y, T = v.shape
        if (T % 2) or not (_fft_len(X) and _fft_len(Y)) or b == 0 or tuple(v_res.shape[:3]) != (b, X, Y) or any(
                p.dtype != torch.float32 for p in reduction.parameters()):
            return None
        lib = _lib.load()
        dev = v.device
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        vc, rc_ = v.detach().contiguous(), v_res.detach().contiguous()
        frames = torch.empty(b, 1, X, Y, T + 1, dtype=torch.float32, device=dev)
        w2t = reduction.weight.detach().reshape(1, C).t().contiguous()
        b2 = reduction.bias.detach().contiguous() if reduction.bias is not None else None
        with torch.cuda.device(dev):
            rc = lib.tcfd_fno_reduce_frames(vc.data_ptr(), frames.data_ptr(), w2t.data_ptr(), b2.data_ptr() if b2 is not None else None,
                                            rc_.data_ptr(), rc_.shape[-1], b, C, X * Y * T, T, stream)
        if rc == -1 and b"not instantiated" in lib.tcfd_last_error():
            return None
        _lib.check(rc, "tcfd_fno_reduce_frames")
        steps = out_steps + 1
        t_pad = (T + 1) if conv.temporal_padding else 0
        vh, plan = hip_truncated_rfftn(frames, conv.modes, t_pad=t_pad, t_out=steps + t_pad, norm=conv.norm)
        oh = hip_contract(vh, list(conv.weight), conv._bias_list(), conv.delta, conv.modes)
        out = torch.empty(b, 1, X, Y, out_steps, dtype=torch.float32, device=dev)
        ws = plan.workspace(b, 1, 1)
        _, inv_scale = _norm_scales(conv.norm, X * Y * (T + 1 + t_pad), X * Y * (steps + t_pad))
        with torch.cuda.device(dev):
            _lib.check(lib.tcfd_fno_inverse_trunc_residual(plan.handle, oh.data_ptr(), out.data_ptr(), rc_.data_ptr(), rc_.shape[-1], b, 1,
                                                           out_steps, inv_scale, ws.data_ptr(), ws.numel(), stream),
                       "tcfd_fno_inverse_trunc_residual")
        out = out.squeeze(1)
        if not training:
            return out
        # training: the same launches as ONE autograd node (the adjoints of the three pieces + the reduction's backward)
        modes = tuple(conv.modes)
        bias = conv._bias_list()
        params = list(conv.weight) + (list(bias) if bias is not None else [])
        cfg = (((b, 1, X, Y, T + 1), modes, t_pad, steps + t_pad, conv.norm),
               (float(conv.delta), modes, True, bias is not None),
               ((X, Y, T + 1, t_pad, steps + t_pad) + modes, out_steps, conv.norm))
        return _OutHeadFn.apply(out, vh, cfg, v, reduction.weight, reduction.bias, *params)

    def forward(self, v, v_res, out_steps: int, **kwargs):
        v_res = v_res.unsqueeze(1).expand(-1, v.size(1), -1, -1, -1)
        v = torch.cat([v_res[..., -1:], v], dim=-1)
        sp = self.spatial_padding
        if sp > 0:
            v = F.pad(v, pad=(0, 0, sp, sp, sp, sp), mode="constant")
        if type(self.conv).forward is SpectralConvT.forward:
            v = self.conv(v, out_steps=out_steps + 1, keep_steps=out_steps)       # the first output step is never formed
        else:
            v = self.conv(v, out_steps=out_steps + 1)[..., -out_steps:]
        if sp > 0:
            v = v[..., sp:-sp, sp:-sp, :]
        if v.is_contiguous() and not (torch.is_grad_enabled() and (v.requires_grad or v_res.requires_grad)):
            v.add_(v_res[..., -1:])         # the convolution's own output buffer: one pass, no second tensor
        else:
            v = v_res[..., -1:] + v
        return v.squeeze(1)


class FNOBase(nn.Module):
    def __init__(self, *, num_spectral_layers: int = 4, fft_norm="backward", activation: ActivationType = "ReLU",
                 spatial_padding: int = 0, channel_expansion: int = 4, spatial_random_feats: bool = False,
                 lift_activation: bool = False, debug=False, **kwargs):
        super().__init__()
        self.spatial_padding = spatial_padding
        self.fft_norm = fft_norm
        self.activation = activation
        self.spatial_random_feats = spatial_random_feats
        self.lift_activation = lift_activation
        self.channel_expansion = channel_expansion
        self.debug = debug
        self.num_spectral_layers = num_spectral_layers

    def double(self):
        """Parameters to float64 / complex128 (fno/base.py:342-349).  An fp64 model runs the SAME fused kernels as an
        fp32 one, instantiated for double (transforms, contraction on v_mfma_f64_16x16x4_f64, pointwise block)."""
        for prm in self.parameters():
            if prm.dtype == torch.float32:
                prm.data = prm.data.to(torch.float64)
            elif prm.dtype == torch.complex64:
                prm.data = prm.data.to(torch.complex128)
        return self


class SFNO(FNOBase):
    """Spectral-refiner FNO for (2+1)-D fields: (b, x, y, t_in) -> (b, x, y, out_steps)."""

    def __init__(self, modes_x: int, modes_y: int, modes_t: int, width: int, out_dim: int = 1, beta: float = -1e-2,
                 delta: float = 1e-1, num_spectral_layers: int = 4, fft_norm: str = "backward",
                 activation: ActivationType = "ReLU", spatial_padding: int = 0, temporal_padding: bool = True,
                 channel_expansion: int = 4, spatial_random_feats: bool = False, lift_activation: bool = True,
                 latent_steps: int = 10, output_steps: int = None, debug=False, **kwargs):
        super().__init__(num_spectral_layers=num_spectral_layers, fft_norm=fft_norm, activation=activation,
                         spatial_padding=spatial_padding, channel_expansion=channel_expansion,
                         spatial_random_feats=spatial_random_feats, lift_activation=lift_activation, debug=debug,
                         **kwargs)
        self.modes_x, self.modes_y, self.modes_t, self.width = modes_x, modes_y, modes_t, width
        if num_spectral_layers < 2:
            raise ValueError("SFNO needs at least two spectral layers (the lifting operator holds the first)")
        # hidden layers: x1 = K(v) ; v <- act(mlp(x1) + w(v)).  The four ModuleList names are the checkpoint
        # contract of the reference (fno/sfno.py:539-556: spectral_conv / mlp / w / activations).
        hidden = range(num_spectral_layers - 1)
        self.spectral_conv = nn.ModuleList(SpectralConvS(width, width, modes_x, modes_y, modes_t) for _ in hidden)
        self.mlp = nn.ModuleList(PointwiseFFN(width, width, channel_expansion * width, activation) for _ in hidden)
        self.w = nn.ModuleList(nn.Conv3d(width, width, 1) for _ in hidden)
        self.activations = nn.ModuleList(getattr(nn, activation)() for _ in hidden)
        self.lifting_operator = LiftingOperator(width, modes_x, modes_y, modes_t, latent_steps=latent_steps,
                                                norm=fft_norm, beta=beta, activation=activation,
                                                spatial_random_feats=spatial_random_feats,
                                                channel_expansion=channel_expansion, nonlinear=lift_activation)
        self.output_operator = OutConv(modes_x, modes_y, modes_t, out_dim=out_dim, delta=delta,
                                       out_steps=output_steps, spatial_padding=spatial_padding,
                                       temporal_padding=temporal_padding, norm=fft_norm)
        self.reduction = nn.Conv3d(width, 1, kernel_size=1)
        self.out_steps = output_steps

    def forward(self, v, out_steps=None):
        if out_steps is None:
            out_steps = self.out_steps if self.out_steps is not None else v.size(-1)
        v_res = v
        v = self.lifting_operator(v.unsqueeze(1))
        for conv, mlp, w, act in zip(self.spectral_conv, self.mlp, self.w, self.activations):
            fused = hip_spectral_layer(conv, v, mlp.linear1, mlp.activation, mlp.linear2, skip_conv=w, act2=act)
            if fused is not None:       # training: the whole layer as one autograd node
                v = fused
                continue
            x1 = conv(v)
            fused = hip_pointwise(x1, mlp.linear1, mlp.activation, mlp.linear2, skip=v, skip_conv=w, act2=act)
            v = fused if fused is not None else act(mlp(x1) + w(v))
        out = self.output_operator.fused_forward(v, v_res, self.reduction, out_steps) if type(self.output_operator) is OutConv else None
        if out is not None:
            return out
        red = hip_pointwise(v, None, None, self.reduction)
        v = red if red is not None else self.reduction(v)
        return self.output_operator(v, v_res, out_steps=out_steps)


# ----------------------------------------------------------------------------- one training iteration as a replayed graph
def make_graphed_training_step(model: nn.Module, loss_fn, optimizer: torch.optim.Optimizer, x: torch.Tensor, y: torch.Tensor,
                               warmup: int = 3):
    """``step(x, y) -> loss`` = zero_grad + forward + loss + backward + optimizer step, captured ONCE in a
    ``torch.cuda.CUDAGraph`` and replayed.  The training loops the reference ships work on small problems (examples/
    ex2_SFNO_train.ipynb: batch 4, 64 x 64 x 10): one iteration is ~240 kernel launches and ~900 tensor-op dispatches, 1.8 ms of
    device time behind 4-5 ms of host time; replayed as a graph it runs at the device's pace.  Every kernel of this package
    is launched on the current stream with caller-owned buffers and never synchronises, so the whole iteration captures.

    ``x`` / ``y`` fix shape and dtype; each call copies its batch into the captured input buffers.  The optimizer must be
    capturable (``torch.optim.Adam(..., capturable=True)``: step counters on the device).  Gradients live in static buffers
    (``zero_grad(set_to_none=False)`` semantics); learning-rate schedulers that write ``param_group["lr"]`` as a Python float
    are frozen at capture time -- keep ``lr`` in a tensor for them."""
    if not x.is_cuda:
        raise _lib.TcfdError("expected HIP device tensors (torch-cfd_amd has no CPU fallback)")
    for group in optimizer.param_groups:
        if "capturable" in group and not group["capturable"]:
            raise ValueError("the optimizer must be created with capturable=True to be replayed in a graph")
    sx, sy = x.detach().clone(), y.detach().clone()

    def iteration():
        optimizer.zero_grad(set_to_none=False)
        loss = loss_fn(model(sx), sy)
        loss.backward()
        optimizer.step()
        return loss

    side = torch.cuda.Stream(device=x.device)
    side.wait_stream(torch.cuda.current_stream(x.device))
    with torch.cuda.stream(side):          # plans, workspaces, lazily built tables, optimizer state: everything exists before capture
        for p_ in model.parameters():
            if p_.grad is None and p_.requires_grad:
                p_.grad = torch.zeros_like(p_)
        for _ in range(max(1, warmup)):
            iteration()
    torch.cuda.current_stream(x.device).wait_stream(side)
    torch.cuda.synchronize(x.device)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = iteration()

    def step(xb: torch.Tensor, yb: torch.Tensor) -> torch.Tensor:
        sx.copy_(xb, non_blocking=True)
        sy.copy_(yb, non_blocking=True)
        graph.replay()
        return static_loss

    step.graph = graph
    return step


# ----------------------------------------------------------------------------- loss: torch-cfd_amd/losses.py (fno/losses.py)
from .losses import SobolevLoss, hip_weighted_sqnorm  # noqa: E402,F401
