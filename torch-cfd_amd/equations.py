"""Operator API of the spectral Navier-Stokes path (drop-in for torch_cfd/equations.py).

Same class names, constructor arguments, method names, ``state_dict`` keys and
error behaviour as the reference:

  * ``stable_time_step``            equations.py:35-64
  * ``ImplicitExplicitODE``         equations.py:67-107
  * ``IMEXStepper``                 equations.py:110-246
  * ``RK4CrankNicolsonStepper``     equations.py:249-358
  * ``NavierStokes2DSpectral``      equations.py:361-463

but ``NavierStokes2DSpectral.forward`` / ``explicit_terms`` / ``residual`` run the
hand-written gfx950 kernels of ``csrc/tcfd_ns2d.hip`` through the C ABI in
``include/tcfd.h`` (three launches per RK stage instead of ~40 ATen launches).
There is no CPU or eager fallback: tensors must live on a HIP device and the grid
must be square with n = 2^k (8..2048), n = 3 * 2^k (96..1536) or n = 5 * 2^k (80..1280) -- the fused kernels -- or another
n = p * 2^k with a small odd p (48, 112, 224, ...: power-of-two HIP transforms + tensor ops, ``mixed_radix.py``)
-- anything else raises.  The fused kernels are
forward-only; when gradients are asked for (a state that requires grad, trainable
stepper coefficients) the operator steps through ``autograd.py``: the same arithmetic
as device tensor ops around the HIP transforms and their hand-written adjoints.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from typing import Callable, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import _lib
from .grids import Grid
from .spectral import brick_wall_filter_2d

Params = Union[nn.ParameterDict, Dict]

_COMPLEX_OF = {torch.float32: torch.complex64, torch.float64: torch.complex128}
_REAL_OF = {torch.complex64: torch.float32, torch.complex128: torch.float64}


def stable_time_step(dx: float = None, dt: float = None, max_velocity: float = 1.0,
                     max_courant_number: float = 0.5, viscosity: float = 1e-3,
                     implicit_diffusion: bool = True, ndim: int = 2) -> float:
    """Largest admissible step: the smallest of the caller's ``dt`` (if any), the advective CFL bound
    ``C dx / v_max`` and the diffusive bound -- ``dx^2 / (2^ndim nu)`` for explicit diffusion, just ``dx`` when
    diffusion is treated implicitly (torch_cfd/equations.py:35-64)."""
    bounds = [max_courant_number * dx / max_velocity,
              dx if implicit_diffusion else dx * dx / (viscosity * 2**ndim)]
    if dt is not None:
        bounds.append(dt)
    return min(bounds)


# ----------------------------------------------------------------------------- HIP plan wrapper
class _HipPlan:
    """Owns one ``tcfd_ns2d_plan`` (device tables) + a workspace cache."""

    def __init__(self, n: int, cdtype: torch.dtype, device: torch.device, kx1d, ky1d, linear_term, mask,
                 forcing_hat=None):
        self.lib = _lib.load()
        self.n, self.m = n, n // 2 + 1
        self.cdtype = cdtype
        self.rdtype = _REAL_OF[cdtype]
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TcfdError("torch-cfd_amd runs on HIP devices only (no CPU fallback); got " + str(device))

        def host64(t):
            return t.detach().to("cpu", torch.float64).contiguous()

        kx1d, ky1d, lin, msk = host64(kx1d), host64(ky1d), host64(linear_term), host64(mask)
        assert kx1d.numel() == n and ky1d.numel() == self.m and lin.shape == (n, self.m) == msk.shape
        fptr = None
        if forcing_hat is not None:
            f = torch.view_as_real(forcing_hat.detach().to("cpu", torch.complex128).contiguous()).contiguous()
            assert f.shape == (n, self.m, 2)
            fptr = _lib.dptr_of_tensor(f)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.tcfd_ns2d_plan_create(
                ctypes.byref(handle), n, _lib.TCFD_C128 if cdtype == torch.complex128 else _lib.TCFD_C64,
                _lib.dptr_of_tensor(kx1d), _lib.dptr_of_tensor(ky1d), _lib.dptr_of_tensor(lin),
                _lib.dptr_of_tensor(msk), fptr)
        _lib.check(rc, "tcfd_ns2d_plan_create")
        self.handle = handle
        self._ws: Optional[torch.Tensor] = None
        self._finalizer = weakref.finalize(self, self.lib.tcfd_ns2d_plan_destroy, handle)

    def info(self) -> Dict[str, int]:
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self.lib.tcfd_ns2d_plan_info(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)),
                   "tcfd_ns2d_plan_info")
        d, e = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self.lib.tcfd_ns2d_plan_variant(self.handle, ctypes.byref(d), ctypes.byref(e)), "tcfd_ns2d_plan_variant")
        return {"separable": a.value, "sparse_forcing": b.value, "keep_cols": c.value, "split": d.value,
                "rows_kernel": e.value}

    # -- helpers
    def workspace(self, batch: int) -> torch.Tensor:
        """Scratch for ``batch`` fields.  One buffer, grown to the largest batch seen: the library carves it by
        ``batch`` alone, so a larger buffer serves smaller batches (alternating batch sizes do not reallocate)."""
        need = self.lib.tcfd_ns2d_workspace_bytes(self.handle, batch)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None   # release before allocating the larger one
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _prep(self, w: torch.Tensor) -> Tuple[torch.Tensor, int]:
        if not w.is_cuda:
            raise _lib.TcfdError("expected a HIP device tensor (torch-cfd_amd has no CPU fallback)")
        if w.device != self.device:
            raise _lib.TcfdError(f"tensor on {w.device}, operator tables on {self.device}")
        # gradients never reach this point: NavierStokes2DSpectral routes them through torch-cfd_amd/autograd.py
        if not w.is_complex() or w.shape[-2:] != (self.n, self.m):
            raise ValueError(f"expected complex (*, {self.n}, {self.m}) half spectrum, got {tuple(w.shape)} {w.dtype}")
        w = w.detach().to(self.cdtype).contiguous()
        batch = w.numel() // (self.n * self.m)
        return w, batch

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- operations
    def step(self, w, beta, gdt, mu, steps: int, inv_total_dt: float, want_dwdt: bool = True, fa=None, mu_den=None,
             base0=None):
        """``steps`` fused IMEX steps; stage k: h <- fa_k F(u) + beta_k h, u <- (base + gdt_k h + mu_k L base) / (1 - mu_den_k L)
        (include/tcfd.h, tcfd_ns2d_step_imex).  fa / mu_den / base0 default to the RK4-CN form (1, mu, current state)."""
        w, batch = self._prep(w)
        out = torch.empty_like(w)
        dwdt = torch.empty_like(w) if want_dwdt else None
        if batch == 0:   # empty batch: the reference's tensor ops return empty tensors
            return out, dwdt
        ws = self.workspace(batch)
        ints = (ctypes.c_int * len(beta))(*[int(v) for v in base0]) if base0 is not None else None
        with torch.cuda.device(self.device):
            rc = self.lib.tcfd_ns2d_step_imex(
                self.handle, w.data_ptr(), out.data_ptr(), dwdt.data_ptr() if want_dwdt else None, batch,
                len(beta), _lib.darray(fa) if fa is not None else None, _lib.darray(beta), _lib.darray(gdt),
                _lib.darray(mu), _lib.darray(mu_den) if mu_den is not None else None, ints, steps, inv_total_dt,
                ws.data_ptr(), ws.numel(), self._stream())
        _lib.check(rc, "tcfd_ns2d_step_imex")
        return out, dwdt

    def explicit_terms(self, w):
        w, batch = self._prep(w)
        out = torch.empty_like(w)
        if batch == 0:
            return out
        ws = self.workspace(batch)
        with torch.cuda.device(self.device):
            rc = self.lib.tcfd_ns2d_explicit_terms(self.handle, w.data_ptr(), out.data_ptr(), batch,
                                                   ws.data_ptr(), ws.numel(), self._stream())
        _lib.check(rc, "tcfd_ns2d_explicit_terms")
        return out

    def explicit_terms_vjp(self, w, gm):
        """The four half spectra X_f of ``tcfd_ns2d_explicit_terms_vjp`` for the state ``w`` and the pre-weighted cotangent
        ``gm`` = mask * g / c: shape (4, *w.shape)."""
        w, batch = self._prep(w)
        gm, _ = self._prep(gm)
        out = torch.empty((4,) + tuple(w.shape), dtype=w.dtype, device=w.device)
        if batch == 0:
            return out
        ws = self.workspace(batch)
        with torch.cuda.device(self.device):
            rc = self.lib.tcfd_ns2d_explicit_terms_vjp(self.handle, w.data_ptr(), gm.data_ptr(), out.data_ptr(), batch,
                                                       ws.data_ptr(), ws.numel(), self._stream())
        _lib.check(rc, "tcfd_ns2d_explicit_terms_vjp")
        return out

    def stream_residual(self, w, wt, want_psi=True, want_res=True):
        w, batch = self._prep(w)
        wt_, _ = self._prep(wt)
        psi = torch.empty_like(w) if want_psi else None
        res = torch.empty_like(w) if want_res else None
        if batch == 0:
            return psi, res
        ws = self.workspace(batch)
        with torch.cuda.device(self.device):
            rc = self.lib.tcfd_ns2d_stream_residual(
                self.handle, w.data_ptr(), wt_.data_ptr(), psi.data_ptr() if want_psi else None,
                res.data_ptr() if want_res else None, batch, ws.data_ptr(), ws.numel(), self._stream())
        _lib.check(rc, "tcfd_ns2d_stream_residual")
        return psi, res

    def velocity(self, w):
        w, batch = self._prep(w)
        uh, vh, psi = torch.empty_like(w), torch.empty_like(w), torch.empty_like(w)
        if batch == 0:
            return (uh, vh), psi
        with torch.cuda.device(self.device):
            rc = self.lib.tcfd_ns2d_velocity(self.handle, w.data_ptr(), uh.data_ptr(), vh.data_ptr(),
                                             psi.data_ptr(), batch, self._stream())
        _lib.check(rc, "tcfd_ns2d_velocity")
        return (uh, vh), psi

    def rfft2(self, x):
        if not x.is_cuda or x.shape[-2:] != (self.n, self.n):
            raise ValueError(f"rfft2 expects a HIP tensor (*, {self.n}, {self.n})")
        x = x.detach().to(self.rdtype).contiguous()
        batch = x.numel() // (self.n * self.n)
        out = torch.empty(*x.shape[:-1], self.m, dtype=self.cdtype, device=x.device)
        if batch == 0:
            return out
        with torch.cuda.device(self.device):
            rc = self.lib.tcfd_rfft2(self.handle, x.data_ptr(), out.data_ptr(), batch, self._stream())
        _lib.check(rc, "tcfd_rfft2")
        return out

    def irfft2(self, xh):
        xh, batch = self._prep(xh)
        out = torch.empty(*xh.shape[:-1], self.n, dtype=self.rdtype, device=xh.device)
        if batch == 0:
            return out
        ws = self.workspace(batch)
        with torch.cuda.device(self.device):
            rc = self.lib.tcfd_irfft2(self.handle, xh.data_ptr(), out.data_ptr(), batch, ws.data_ptr(),
                                      ws.numel(), self._stream())
        _lib.check(rc, "tcfd_irfft2")
        return out

    def subsample_factor(self, out_size) -> int:
        """The factor ``irfft2_subsample`` runs ``n -> out_size`` with, or 0 when only the two-step path covers it."""
        if not out_size or out_size >= self.n or self.n % out_size:
            return 0
        f = self.n // out_size
        # the kernel's own bound (the lanes of one row transform: 16 at n = 256, 8 at n = 64, ...); above it the two calls run
        return f if (f & (f - 1)) == 0 and f <= self.lib.tcfd_irfft2_subsample_max_factor(self.handle) else 0

    def irfft2_subsample(self, xh, factor: int):
        """``F.interpolate(irfft2(xh), size=(n / factor,) * 2, mode="bilinear")`` as one pass (tcfd_irfft2_subsample)."""
        xh, batch = self._prep(xh)
        ns = self.n // factor
        out = torch.empty(*xh.shape[:-2], ns, ns, dtype=self.rdtype, device=xh.device)
        if batch == 0:
            return out
        ws = self.workspace(batch)
        with torch.cuda.device(self.device):
            rc = self.lib.tcfd_irfft2_subsample(self.handle, xh.data_ptr(), out.data_ptr(), batch, factor, ws.data_ptr(),
                                                ws.numel(), self._stream())
        _lib.check(rc, "tcfd_irfft2_subsample")
        return out


_MESH_PLANS: Dict[tuple, object] = {}


class _MeshTables:
    """The table attributes the tensor-op plan reads from an operator, for the stand-alone helpers."""

    smooth = False

    def __init__(self, kx, ky):
        self.kx, self.ky = kx, ky
        self.linear_term = torch.zeros_like(kx)
        self.filter = torch.ones_like(kx)


def _composite_plan(op, n: int, cdtype, device, forcing_hat=None):
    """Plan of a grid outside the fused kernels' sizes: n = p * 2^k with a small odd p on power-of-two HIP transforms, any
    other even n on dense device transforms; the stage loop in tensor ops either way (mixed_radix.py)."""
    from .mixed_radix import CompositeFft, DenseDft, TensorOpPlan, odd_factor_split

    split = odd_factor_split(n)
    if split is not None:
        p_, m_ = split
        return TensorOpPlan(op, CompositeFft(n, p_, fft_plan(m_, cdtype, device)), device, forcing_hat)
    if n % 2 or n < 4 or n > 4096:
        raise _lib.TcfdError(f"n = {n}: the HIP spectral path covers even n (the reference's irfft2 default size, "
                             "torch_cfd/equations.py:413-422): 2^k, 3 * 2^k, 5 * 2^k on the fused kernels, the rest up to 4096 "
                             "through composite / dense transforms")
    return TensorOpPlan(op, DenseDft(n, cdtype), device, forcing_hat)


def _is_pow2(n: int) -> bool:
    """Grids the FUSED kernels cover: n = 2^k (8..2048), n = 3 * 2^k (96..1536: radix-12 first pass) and n = 5 * 2^k
    (80..1280: radix 20)."""
    return (8 <= n <= 2048 and (n & (n - 1)) == 0) or n in (96, 192, 384, 768, 1536, 80, 160, 320, 640, 1280)


def _plan_for_mesh(kx: torch.Tensor, ky: torch.Tensor, like: torch.Tensor):
    """Table-free plan (L = 0, mask = 1) keyed on the mesh, for the stand-alone
    helpers (vorticity_to_velocity, rfft2/irfft2)."""
    n, m = kx.shape[-2:]
    cdtype = torch.promote_types(like.dtype, _COMPLEX_OF.get(kx.dtype, torch.complex64))
    kx1 = kx[..., :, 0].detach().to("cpu", torch.float64).contiguous()
    ky1 = ky[..., 0, :].detach().to("cpu", torch.float64).contiguous()
    key = (n, cdtype, like.device, float(kx1[1]), float(ky1[1]))
    plan = _MESH_PLANS.get(key)
    if plan is None:
        if _is_pow2(n):
            plan = _HipPlan(n, cdtype, like.device, kx1, ky1, torch.zeros(n, m), torch.ones(n, m))
        else:
            real = _REAL_OF[cdtype]
            plan = _composite_plan(_MeshTables(kx.detach().to(like.device, real), ky.detach().to(like.device, real)), n, cdtype,
                                   like.device)
        _MESH_PLANS[key] = plan
    return plan


def fft_plan(n: int, cdtype: torch.dtype, device, diam: float = 2 * torch.pi) -> _HipPlan:
    """Plan for plain rfft2/irfft2 of (*, n, n) fields on ``device``."""
    k = torch.fft.fftfreq(n, d=diam / n, dtype=_REAL_OF[cdtype])
    kx, ky = torch.meshgrid(k, k, indexing="ij")
    like = torch.empty(0, dtype=cdtype, device=device)
    return _plan_for_mesh(kx[:, : n // 2 + 1], ky[:, : n // 2 + 1], like)


# ----------------------------------------------------------------------------- ODE interface + steppers
class ImplicitExplicitODE(nn.Module):
    r"""du/dt = explicit_terms(u) + implicit_terms(u) with a linear implicit part that ``implicit_solve`` inverts:
    ``implicit_solve(u, s)`` returns ``(1 - s L)^{-1} u``.  (Interface of torch_cfd/equations.py:67-107.)"""

    def explicit_terms(self, u):
        raise NotImplementedError

    def implicit_terms(self, u):
        raise NotImplementedError

    def implicit_solve(self, u: torch.Tensor, step_size: float):
        raise NotImplementedError

    def residual(self, u: torch.Tensor, u_t: torch.Tensor):
        raise NotImplementedError


def run_stage_schedule(u: torch.Tensor, equation: ImplicitExplicitODE, sched: Dict[str, list]) -> torch.Tensor:
    """One IMEX step of ANY ``ImplicitExplicitODE`` from the per-stage scalars the fused HIP step consumes
    (include/tcfd.h, ``tcfd_ns2d_step_imex``).  Stage k:

        h <- fa_k F(u) + beta_k h
        u <- (1 - mu_den_k L)^{-1} (b + gdt_k h + mu_k L b),    b = u, or the step's initial state when base0_k

    Every scheme of this module (forward/backward Euler, IMEX-CN, RK2-CN, low-storage RK-CN) is an instance, so this
    one loop is the generic (non-fused) form of all of them; on a ``NavierStokes2DSpectral`` the same scalars go to
    the kernels instead and this function is only the cross-check (``tests/test_ns2d_gpu.py``)."""
    n = len(sched["beta"])
    fa = sched.get("fa") or [1.0] * n
    mu_den = sched.get("mu_den") or sched["mu"]
    base0 = sched.get("base0") or [0] * n
    start, h = u, None
    for k in range(n):
        f = equation.explicit_terms(u)
        h = fa[k] * f if h is None else fa[k] * f + sched["beta"][k] * h
        b = start if base0[k] else u
        u = equation.implicit_solve(b + sched["gdt"][k] * h + sched["mu"][k] * equation.implicit_terms(b), mu_den[k])
    return u


class IMEXStepper(nn.Module):
    """One-step IMEX schemes: ``order`` 1 / 1.5 (explicit Euler + theta-weighted implicit part, theta = ``alpha``) and
    2 (two-stage RK + Crank-Nicolson).  A scheme is DATA here -- ``stage_schedule`` turns the parameters and ``dt`` into
    per-stage scalars -- executed either by the fused HIP kernels (equation = ``NavierStokes2DSpectral``) or by
    ``run_stage_schedule`` (any other equation).  Constructor / ``params`` names follow torch_cfd/equations.py:110-246."""

    def __init__(self, order: float = 2, alpha: float = 0.5, beta: Optional[float] = 0.5,
                 requires_grad: bool = False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if order not in (1, 1.5, 2, 4) or (order == 4 and type(self) is IMEXStepper):
            # order 4 is RK4CrankNicolsonStepper's; the reference's base class would be left without a stepper
            raise ValueError(f"IMEXStepper: unsupported order {order}")
        self.order = order
        self._set_params({"alpha": torch.tensor(alpha), "beta": torch.tensor(beta)}, requires_grad=requires_grad)

    def _set_params(self, params: Params, requires_grad: bool = False):
        self.params = nn.ParameterDict({k: nn.Parameter(v, requires_grad=requires_grad) for k, v in params.items()})
        self.requires_grad = requires_grad

    def stage_schedule(self, params: Params, dt: float, as_tensors: bool = False) -> Dict[str, list]:
        """Per-stage scalars, each rounded the way the reference's 0-dim tensor arithmetic rounds it
        (``(1 - alpha) * dt`` is a default-dtype tensor product there, equations.py:174-228).  ``as_tensors`` keeps
        the entries that depend on ``alpha`` / ``beta`` as 0-dim tensors attached to the parameters, so a trainable
        scheme (``requires_grad=True``) receives gradients when the step runs in its differentiable form."""
        if as_tensors:
            alpha, num = params["alpha"], (lambda t: t)
        else:
            alpha, num = params["alpha"].detach().cpu(), (lambda t: t.item())
        if self.order in (1, 1.5):
            return {"fa": [1.0], "beta": [0.0], "gdt": [float(dt)], "mu": [num((1 - alpha) * dt)],
                    "mu_den": [num(alpha * dt)], "base0": [0]}
        half = num((params["beta"] if as_tensors else params["beta"].detach().cpu()) * dt)   # Crank-Nicolson weight of both stages
        return {"fa": [1.0, num(alpha * 1)], "beta": [0.0, num(1 - alpha)], "gdt": [float(dt)] * 2,
                "mu": [half, half], "mu_den": [half, half], "base0": [0, 1]}

    def stepper(self, u, dt, equation, params=None):
        """The step in its generic form (explicit terms of ``equation`` + element-wise tensor ops)."""
        return run_stage_schedule(u, equation, self.stage_schedule(self.params if params is None else params, dt))

    def forward(self, u, dt, equation, params=None):
        params = self.params if params is None else params
        if isinstance(equation, NavierStokes2DSpectral) and _is_module_stepper(self):
            out, _ = equation._fused_steps(u, dt, 1, params, want_dwdt=False, stepper=self)
            return out
        return self.stepper(u, dt, equation, params)


def _is_module_stepper(solver) -> bool:
    """True when ``solver`` steps exactly as this module's schemes do, i.e. its stage schedule may go to the fused
    kernels.  A user subclass that overrides ``forward`` / ``stepper`` / ``stage_schedule`` is a different scheme:
    it is called as ``solver(u, dt, equation)`` (its explicit terms still run on HIP), never replaced silently."""
    if not isinstance(solver, IMEXStepper):
        return False
    cls = type(solver)
    own_schedules = (IMEXStepper.stage_schedule, RK4CrankNicolsonStepper.stage_schedule)
    return (cls.forward is IMEXStepper.forward and cls.stepper is IMEXStepper.stepper
            and cls.stage_schedule in own_schedules and "stepper" not in vars(solver) and "forward" not in vars(solver))


_CARPENTER_KENNEDY = {   # 5-stage low-storage RK4 (2N storage), the reference's default weights (equations.py:294-317)
    "alphas": [0, 0.1496590219993, 0.3704009573644, 0.6222557631345, 0.9582821306748, 1],
    "betas": [0, -0.4178904745, -1.192151694643, -1.697784692471, -1.514183444257],
    "gammas": [0.1496590219993, 0.3792103129999, 0.8229550293869, 0.6994504559488, 0.1530572479681],
}
_CLASSIC_RK4 = {"alphas": [0.0, 0.5, 0.5, 1.0, 1.0], "betas": [0.0, 0.0, 0.0, 0.0], "gammas": [1 / 6, 1 / 3, 1 / 3, 1 / 6]}


class RK4CrankNicolsonStepper(IMEXStepper):
    """Low-storage Runge-Kutta for the explicit part, Crank-Nicolson over each stage interval for the implicit one:
    ``mu_k = dt/2 (alpha_{k+1} - alpha_k)``.  ``params`` holds ``alphas / betas / gammas`` in the default dtype at
    construction, as in the reference (a float32 run steps with float32-rounded coefficients).  ``weights`` replaces
    the tables; ``low_storage=False`` selects classic RK4 weights (stored as floats: the reference builds an integer
    tensor for that case and cannot construct the module)."""

    def __init__(self, order: float = 4, requires_grad: bool = False, weights: Optional[Params] = None,
                 low_storage: bool = True, *args, **kwargs):
        super().__init__(order, *args, **kwargs)
        table = weights if weights is not None else (_CARPENTER_KENNEDY if low_storage else _CLASSIC_RK4)
        dtype = torch.get_default_dtype()
        self._set_params({k: torch.as_tensor(v, dtype=dtype).clone() for k, v in table.items()}, requires_grad=requires_grad)

    @staticmethod
    def stage_scalars(params: Params, dt: float):
        """(beta_k, gamma_k dt, mu_k) as Python floats, rounded like the reference's 0-dim tensor products."""
        al, be, ga = (params[k].detach().cpu() for k in ("alphas", "betas", "gammas"))
        if len(al) - 1 != len(be) != len(ga):   # the reference's own (chained) comparison, equations.py:350
            raise ValueError("number of RK coefficients does not match")
        stages = range(len(be))
        return ([be[k].item() for k in stages], [(ga[k] * dt).item() for k in stages],
                [(0.5 * dt * (al[k + 1] - al[k])).item() for k in stages])

    def stage_schedule(self, params: Params, dt: float, as_tensors: bool = False) -> Dict[str, list]:
        # (trainable RK coefficients do not come through here: autograd.rk_crank_nicolson_steps reads the tensors)
        beta, gdt, mu = self.stage_scalars(params, dt)
        return {"beta": beta, "gdt": gdt, "mu": mu, "fa": None, "mu_den": None, "base0": None}


# ----------------------------------------------------------------------------- the operator
class NavierStokes2DSpectral(ImplicitExplicitODE):
    """2-D vorticity equation on a periodic box, pseudo-spectral:

        dw/dt = -(u . grad) w [+ f]      explicit, 2/3-rule de-aliased       (``explicit_terms``)
              + (nu lap - drag) w        implicit                            (``implicit_terms / implicit_solve``)

    Constructor, buffers (``kx ky laplace linear_term filter`` -- the ``state_dict`` contract) and methods as in
    torch_cfd/equations.py:361-463; every method that touches the state runs the HIP kernels."""

    #: Entries of the forcing spectrum below ``forcing_noise_floor * eps * max|f^|`` are set to exact zeros: they are
    #: the round-off of transforming an analytically band-limited forcing (sin(k y) leaves ~2e-14 relative noise at
    #: n = 1024 in fp64), carry no information, and exact zeros let the plan use its sparse / pruned forms.
    #: ``None`` -> n (the grid size); 0 keeps every entry.  A caller-visible difference from the reference, see
    #: INTEGRATION.md.
    forcing_noise_floor: Optional[float] = None

    def __init__(self, viscosity: float, grid: Grid, drag: float = 0.0, smooth: bool = True,
                 forcing_fn: Optional[Callable] = None, solver: IMEXStepper = None, **kwargs):
        super().__init__()
        self.viscosity, self.grid, self.drag, self.smooth = viscosity, grid, drag, smooth
        self.forcing_fn = forcing_fn
        self.solver = solver
        self._plans: Dict[tuple, _HipPlan] = {}
        self._coef_cache = None
        self._initialize()

    def _initialize(self):
        kx, ky = self.grid.rfft_mesh()
        laplace = -4 * torch.pi**2 * (abs(kx) ** 2 + abs(ky) ** 2)
        for name, table in (("kx", kx), ("ky", ky), ("laplace", laplace),
                            ("linear_term", self.viscosity * laplace - self.drag),
                            ("filter", brick_wall_filter_2d(self.grid))):
            self.register_buffer(name, table)

    def __getstate__(self):  # device plans are rebuilt lazily after copy / unpickle
        state = self.__dict__.copy()
        state["_plans"], state["_coef_cache"] = {}, None
        return state

    # -- forcing table: sampled once per plan
    def forcing_hat(self) -> Optional[torch.Tensor]:
        """(n, m) spectrum of the force on the vorticity equation: rfft2 of the sampled forcing (its curl for a
        momentum forcing), small entries flushed to zero (``forcing_noise_floor``).  Evaluated on the CPU in the
        precision of the operator's tables, once per plan."""
        if self.forcing_fn is None:
            return None
        real = self.kx.dtype
        try:
            sampled = self.forcing_fn(self.grid, None)
        except Exception as e:  # a forcing that needs the state cannot be a plan table
            raise _lib.TcfdError(
                "forcing_fn(grid, None) failed: the HIP spectral path supports state-independent forcings only "
                f"(sampled once per plan) -- {type(e).__name__}: {e}") from e

        def spectrum(field):
            return torch.fft.rfft2(field.data.detach().to("cpu", real))

        if getattr(self.forcing_fn, "vorticity", False):
            fh = spectrum(sampled)
        else:
            fxh, fyh = (spectrum(c) for c in sampled)
            kx, ky = self.kx.detach().cpu(), self.ky.detach().cpu()
            fh = 2j * torch.pi * (fyh * kx - fxh * ky)   # curl in k-space
        scale = self.kx.shape[-2] if self.forcing_noise_floor is None else self.forcing_noise_floor
        if scale > 0:
            mag = fh.abs()
            fh = torch.where(mag > scale * torch.finfo(real).eps * mag.max(), fh, torch.zeros_like(fh))
        return fh

    def _forcing_key(self):
        fn = self.forcing_fn
        if fn is None:
            return None
        return fn.fingerprint() if hasattr(fn, "fingerprint") else ("id", id(fn))

    def invalidate_plan(self):
        """Drop the device plans (tables, forcing spectrum): the next call rebuilds them.  Needed only after
        mutating a user-defined forcing that has no ``fingerprint()``; buffer and built-in forcing changes are
        detected."""
        self._plans, self._coef_cache = {}, None

    def _plan(self, like: torch.Tensor) -> _HipPlan:
        cdtype = torch.promote_types(like.dtype, _COMPLEX_OF.get(self.linear_term.dtype, torch.complex64))
        tables = (self.kx, self.ky, self.linear_term, self.filter)
        key = (cdtype, like.device, self.smooth, self._forcing_key(), self.forcing_noise_floor) + tuple(
            (t.data_ptr(), t._version) for t in tables)
        plan = self._plans.get(key)
        if plan is None:
            n, m = self.kx.shape[-2:]
            if self.grid.shape[0] != self.grid.shape[1] or n != self.grid.shape[0]:
                raise ValueError("the HIP spectral path needs a square n x n grid")
            mask = self.filter if self.smooth else torch.ones_like(self.filter)
            if _is_pow2(n):
                plan = _HipPlan(n, cdtype, like.device, self.kx[:, 0], self.ky[0, :], self.linear_term, mask, self.forcing_hat())
            else:   # n = p * 2^k: power-of-two HIP transforms + the stage loop in tensor ops (mixed_radix.py)
                if not like.is_cuda:
                    raise _lib.TcfdError("expected a HIP device tensor (torch-cfd_amd has no CPU fallback)")
                plan = _composite_plan(self, n, cdtype, like.device, self.forcing_hat())
            self._plans = {key: plan}  # tables changed -> drop stale plans
        return plan

    # -- gradients: the differentiable form of the same arithmetic (autograd.py), only when somebody asks
    @staticmethod
    def _wants_grad(*tensors) -> bool:
        return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)

    def _forcing_on(self, like: torch.Tensor):
        """The forcing spectrum on the device, sampled once per plan (the plan is rebuilt when the forcing's fingerprint or a
        table changes): sampling it again on every differentiable call cost a CPU rfft2 of the grid each time (30-50 ms at
        1024^2 -- more than the step)."""
        plan = self._plan(like)
        cached = getattr(plan, "_forcing_dev", None)
        if cached is None:
            fh = self.forcing_hat()
            cached = (None if fh is None else fh.to(device=like.device, dtype=plan.cdtype),)
            plan._forcing_dev = cached
        return cached[0]

    def _autograd_steps(self, vort_hat, dt, steps, params, stepper, want_dwdt):
        from . import autograd as ad

        plan = self._plan(vort_hat)
        w = vort_hat.to(plan.cdtype)
        lead = w.shape
        w = w.reshape(-1, plan.n, plan.m)
        forcing = self._forcing_on(w)
        constant = not any(torch.is_tensor(v) and v.requires_grad for v in params.values())
        if (constant and ad._fused_vjp_ok(plan) and _is_module_stepper(stepper) and os.environ.get("TCFD_FUSED_STAGE", "1") != "0"
                and w.is_cuda):
            # only the STATE asks for gradients: the schedule is plain numbers (rounded as the fused forward kernels round
            # them), every stage is two fused nodes -- F(w) with its VJP and the stage update with its VJP
            out = ad.fused_scheduled_steps(self, plan, w, steps, stepper.stage_schedule(params, dt), forcing)
        elif isinstance(stepper, RK4CrankNicolsonStepper):
            out = ad.rk_crank_nicolson_steps(self, plan, w, dt, steps, params, forcing)
        else:   # alpha / beta stay tensors: a trainable IMEX scheme gets its gradients (as the reference's 0-dim arithmetic gives)
            sched = stepper.stage_schedule(params, dt, as_tensors=True)
            sched = {k: [v.to(w.device) if torch.is_tensor(v) else v for v in vals] for k, vals in sched.items()}
            out = ad.scheduled_steps(self, plan, w, steps, sched, forcing)
        out = out.reshape(lead)
        return out, ((out - vort_hat) / (steps * dt) if want_dwdt else None)

    def _fused_steps(self, vort_hat, dt, steps, params=None, want_dwdt=True, stepper=None):
        stepper = self.solver if stepper is None else stepper
        if stepper is None:
            raise TypeError("NavierStokes2DSpectral needs a solver (e.g. RK4CrankNicolsonStepper()) to step")
        params = stepper.params if params is None else params
        if self._wants_grad(vort_hat, *params.values()):
            return self._autograd_steps(vort_hat, dt, steps, params, stepper, want_dwdt)
        # the coefficient tensors may live on the GPU: read them back once, not per step
        ckey = (float(dt), id(stepper)) + tuple((k, v.data_ptr(), v._version) for k, v in params.items())
        if self._coef_cache is None or self._coef_cache[0] != ckey:
            self._coef_cache = (ckey, stepper.stage_schedule(params, dt))
        sc = self._coef_cache[1]
        out, dwdt = self._plan(vort_hat).step(vort_hat, sc["beta"], sc["gdt"], sc["mu"], steps, 1 / (steps * dt), want_dwdt,
                                              fa=sc["fa"], mu_den=sc["mu_den"], base0=sc["base0"])
        return out.reshape(vort_hat.shape), (dwdt.reshape(vort_hat.shape) if want_dwdt else None)

    # -- ImplicitExplicitODE interface
    def explicit_terms(self, vort_hat):
        if self._wants_grad(vort_hat):
            from . import autograd as ad

            plan = self._plan(vort_hat)
            w = vort_hat.to(plan.cdtype).reshape(-1, plan.n, plan.m)
            return ad.explicit_terms(self, plan, w, self._forcing_on(w)).reshape(vort_hat.shape)
        return self._plan(vort_hat).explicit_terms(vort_hat).reshape(vort_hat.shape)

    _explicit_terms = explicit_terms

    def implicit_terms(self, vort_hat):
        return self.linear_term * vort_hat

    def implicit_solve(self, vort_hat, dt):
        return vort_hat / (1 - dt * self.linear_term)

    def residual(self, vhat: torch.Tensor, vt_hat: torch.Tensor):
        if self._wants_grad(vhat, vt_hat):
            return vt_hat - self.explicit_terms(vhat) - self.implicit_terms(vhat)
        _, res = self._plan(vhat).stream_residual(vhat, vt_hat, want_psi=False)
        return res.reshape(vhat.shape)

    def stream_and_residual(self, vhat: torch.Tensor, vt_hat: torch.Tensor):
        """(psi_hat, residual) in one fused sweep -- what the trajectory recorder needs per record."""
        psi, res = self._plan(vhat).stream_residual(vhat, vt_hat)
        return psi.reshape(vhat.shape), res.reshape(vhat.shape)

    # -- time stepping
    def forward(self, vort_hat, dt, steps=1) -> Tuple[torch.Tensor, torch.Tensor]:
        """``vort_hat``: (B, n, m), (B, T, n, m) or (n, m) half spectrum.  Returns the state after ``steps`` steps and
        ``(new - old) / (steps * dt)``.  With a stepper of this module all stages of all steps run fused in the HIP
        kernels; a foreign ``solver(u, dt, equation)`` callable is looped over (its explicit terms still run on HIP)."""
        if _is_module_stepper(self.solver):
            return self._fused_steps(vort_hat, dt, steps)
        if self.solver is None:
            raise TypeError("NavierStokes2DSpectral.forward needs a solver (e.g. RK4CrankNicolsonStepper())")
        new = vort_hat
        for _ in range(steps):
            new = self.solver(new, dt, self)
        return new, (new - vort_hat) / (steps * dt)

    step = forward
