"""Differentiable form of the spectral step: the HIP transforms as ``autograd.Function``s + the stage loop in tensor ops.

The fused step kernels are forward-only.  When gradients are asked for -- the state requires grad, or the stepper's
coefficients are trainable (the reference allows both: torch_cfd/equations.py:139, 285;
fno/data_gen/solvers.py:199) -- ``NavierStokes2DSpectral`` steps through this module instead: the same arithmetic as
the kernels, written as device tensor operations around the two hand-written transforms, whose adjoints are again the
hand-written transforms:

    y = irfft2(X)     (c2r drops Im of the DC / Nyquist columns)     grad X = (c / n^2) rfft2(grad y)
    X = rfft2(y)                                                     grad y = n^2 irfft2(grad X / c)

with c = 1 on the DC and Nyquist columns of the half spectrum and 2 elsewhere (the c2r transform counts the interior
columns twice, the r2c transform once).  Nothing here touches torch.fft or the CPU.  Since round 3 the explicit terms and
their vector-Jacobian product run on the fused kernels (``FusedExplicitTerms``: 3 launches forward, ~10 backward), and with
constant coefficients the Runge-Kutta / Crank-Nicolson bookkeeping of a stage is one launch each way as well
(``StageUpdate``); the tensor-op stage loops below remain for trainable coefficients (whose gradients they produce), for the
composite grids and as the cross-check (``TCFD_FUSED_STAGE=0``, ``TCFD_FUSED_VJP=0``).
"""
from __future__ import annotations

import torch


def _column_weights(plan, like: torch.Tensor) -> torch.Tensor:
    """1 on the DC / Nyquist columns of the half spectrum, 2 elsewhere; built once per (plan, device) -- writing the two ones in
    place cost a blocking host-to-device copy per call (and cannot be captured in a graph)."""
    cached = getattr(plan, "_col_weights", None)
    if cached is None or cached.device != like.device:
        host = [2.0] * plan.m
        host[0] = host[-1] = 1.0
        cached = plan._col_weights = torch.tensor(host, dtype=plan.rdtype).to(like.device)
    return cached


class Rfft2(torch.autograd.Function):
    """(*, n, n) real -> (*, n, m) half spectrum on the HIP kernels."""

    @staticmethod
    def forward(ctx, y, plan):
        ctx.plan = plan
        return plan.rfft2(y)

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        scale = float(plan.n * plan.n) / _column_weights(plan, g)
        return Irfft2.apply(g * scale, plan), None


class Irfft2(torch.autograd.Function):
    """(*, n, m) half spectrum -> (*, n, n) real on the HIP kernels (unnormalised inverse / n^2, as torch's)."""

    @staticmethod
    def forward(ctx, xh, plan):
        ctx.plan = plan
        return plan.irfft2(xh)

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        scale = _column_weights(plan, g) / float(plan.n * plan.n)
        return Rfft2.apply(g, plan) * scale, None


class FusedExplicitTerms(torch.autograd.Function):
    """F(w) on the fused kernels (3 launches) with its vector-Jacobian product on the fused kernels too
    (``tcfd_ns2d_explicit_terms_vjp``: the opening column pass of the forward + one column inverse transform, ONE row pass with
    five c2r / four products / four r2c per row pair, four column transforms) instead of ~25 tensor-op launches each way.

    F(w) = M . R(-(I(a2 w) I(a0 w) + I(a3 w) I(a1 w))) + f with a0 = -2 pi i ky / lap (u), a1 = 2 pi i kx / lap (v),
    a2 = 2 pi i kx (dx w), a3 = 2 pi i ky (dy w).  With the transform adjoints of this module (R^T g = n^2 I(g / c),
    I^T y = (c / n^2) R(y)):  nbar = R^T(M g) and  wbar = -(c / n^2) sum_f conj(a_f) R(nbar . partner_f)."""

    @staticmethod
    def forward(ctx, w_hat, op, plan):
        ctx.op, ctx.plan = op, plan
        ctx.save_for_backward(w_hat)
        return plan.explicit_terms(w_hat).reshape(w_hat.shape)

    @staticmethod
    def _tables(op, plan, device):
        key = (plan.cdtype, device, op.smooth) + tuple((t.data_ptr(), t._version) for t in (op.kx, op.ky, op.filter))
        cached = getattr(plan, "_vjp_tables", None)
        if cached is None or cached[0] != key:
            real = plan.rdtype
            kx, ky = op.kx.to(device=device, dtype=real), op.ky.to(device=device, dtype=real)
            lap = (-4 * torch.pi**2 * (kx**2 + ky**2)).clone()
            lap[..., 0, 0] = 1
            two_pi_i = 2j * torch.pi
            a = torch.stack([-two_pi_i * ky / lap, two_pi_i * kx / lap, two_pi_i * kx, two_pi_i * ky * torch.ones_like(kx)]).to(plan.cdtype)
            c = _column_weights(plan, kx)
            mask = op.filter.to(device=device, dtype=real) if op.smooth else torch.ones_like(kx)
            pre = (mask / c).to(real)                                   # gm = g * mask / c
            post = (-(c / float(plan.n * plan.n)) * a.conj()).contiguous()   # (4, n, m): wbar = sum_f post_f X_f
            cached = (key, pre, post)
            plan._vjp_tables = cached
        return cached[1], cached[2]

    @staticmethod
    def backward(ctx, g):
        (w_hat,) = ctx.saved_tensors
        if torch.is_grad_enabled():
            # create_graph=True (gradient penalties, Hessian-vector products): the raw-pointer launches below would cut
            # the VJP's own dependence on w and g out of the graph.  Form it from the differentiable tensor-op restatement
            # (Rfft2 / Irfft2 differentiate through each other any number of times), as the reference's pure-torch graph allows.
            with torch.enable_grad():
                w = w_hat if w_hat.requires_grad else w_hat.detach().requires_grad_(True)
                f = _explicit_terms_tensor_ops(ctx.op, ctx.plan, w, None)
                (wbar,) = torch.autograd.grad(f, w, g.reshape(f.shape), create_graph=True)
            return wbar, None, None
        pre, post = FusedExplicitTerms._tables(ctx.op, ctx.plan, g.device)
        lead = w_hat.shape
        w3 = w_hat.reshape(-1, ctx.plan.n, ctx.plan.m)
        gm = (g.reshape(w3.shape) * pre).contiguous()
        X = ctx.plan.explicit_terms_vjp(w3, gm).contiguous()             # (4, B, n, m)
        wbar = torch.empty_like(gm)
        import ctypes

        from . import _lib

        plane = ctx.plan.n * ctx.plan.m
        with torch.cuda.device(g.device):
            rc = _lib.load().tcfd_ns2d_vjp_combine(X.data_ptr(), post.data_ptr(), wbar.data_ptr(), wbar.numel() // plane, plane,
                                                   _lib.TCFD_C128 if wbar.dtype == torch.complex128 else _lib.TCFD_C64,
                                                   ctypes.c_void_p(torch.cuda.current_stream(g.device).cuda_stream))
        _lib.check(rc, "tcfd_ns2d_vjp_combine")
        return wbar.reshape(lead), None, None


def _fused_vjp_ok(plan) -> bool:
    import os

    return hasattr(plan, "explicit_terms_vjp") and os.environ.get("TCFD_FUSED_VJP", "1") != "0"


def explicit_terms(op, plan, w_hat: torch.Tensor, forcing_hat) -> torch.Tensor:
    """F(w) = mask * rfft2(-(dx w * u + dy w * v)) + f^  with  psi = -w / lap,  (u, v) = (dy psi, -dx psi).  On the fused
    grids (power-of-two, 3 * 2^k, 5 * 2^k plans) through ``FusedExplicitTerms``; the tensor-op form below serves the composite
    grids and as the cross-check (``TCFD_FUSED_VJP=0``)."""
    if _fused_vjp_ok(plan):
        if torch.is_grad_enabled() and w_hat.requires_grad:
            return FusedExplicitTerms.apply(w_hat, op, plan)
        return plan.explicit_terms(w_hat).reshape(w_hat.shape)     # nothing to differentiate: the plain fused sweep
    return _explicit_terms_tensor_ops(op, plan, w_hat, forcing_hat)


def _explicit_terms_tensor_ops(op, plan, w_hat: torch.Tensor, forcing_hat) -> torch.Tensor:
    """The same F(w) as device tensor operations around the two differentiable transforms (any order of derivative)."""
    kx, ky = op.kx.to(w_hat.device), op.ky.to(w_hat.device)
    two_pi_i = 2j * torch.pi
    lap = -4 * torch.pi**2 * (kx**2 + ky**2)
    lap = lap.clone()
    lap[..., 0, 0] = 1
    psi = -w_hat / lap
    fields = (two_pi_i * ky * psi, -two_pi_i * kx * psi, two_pi_i * kx * w_hat, two_pi_i * ky * w_hat)
    u, v, wx, wy = (Irfft2.apply(f.to(plan.cdtype), plan) for f in fields)
    out = Rfft2.apply(-(wx * u + wy * v), plan)
    if op.smooth:
        out = out * op.filter.to(w_hat.device)
    if forcing_hat is not None:
        out = out + forcing_hat
    return out


def rk_crank_nicolson_steps(op, plan, w_hat, dt, steps, params, forcing_hat):
    """``steps`` low-storage RK + Crank-Nicolson steps with the coefficient TENSORS of ``params`` (alphas / betas /
    gammas), so that trainable coefficients receive gradients."""
    lin = op.linear_term.to(w_hat.device)
    al, be, ga = (params[k].to(w_hat.device) for k in ("alphas", "betas", "gammas"))
    u = w_hat
    for _ in range(steps):
        h = None
        for k in range(len(be)):
            f = explicit_terms(op, plan, u, forcing_hat)
            h = f if h is None else f + be[k] * h
            mu = 0.5 * dt * (al[k + 1] - al[k])
            u = (u + ga[k] * dt * h + mu * lin * u) / (1 - mu * lin)
    return u


class StageUpdate(torch.autograd.Function):
    """One stage of the low-storage RK / Crank-Nicolson schedule with CONSTANT coefficients,
        h = fa f + beta h_prev ,   u = (b + gdt h + mu L b) / (1 - mud L)        (the fused forward kernels' own arithmetic),
    and its vector-Jacobian product, one launch each (``tcfd_ns2d_stage_update[_vjp]``) instead of ~10 / ~20 element-wise
    tensor launches.  Linear, so nothing is saved for the backward but the table of L and five numbers."""

    @staticmethod
    def _launch(name, ptrs_in, lin, coef, ptrs_out, like):
        import ctypes

        from . import _lib

        plane = lin.numel()
        code = _lib.TCFD_C128 if like.dtype == torch.complex128 else _lib.TCFD_C64
        c = (ctypes.c_double * 5)(*coef)
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(like.device):
            rc = getattr(_lib.load(), name)(*[ptr(t) for t in ptrs_in], lin.data_ptr(), c, *[ptr(t) for t in ptrs_out],
                                            like.numel() // plane, plane, code,
                                            ctypes.c_void_p(torch.cuda.current_stream(like.device).cuda_stream))
        _lib.check(rc, name)

    @staticmethod
    def forward(ctx, f, h_prev, b, lin, coef):
        f, b = f.contiguous(), b.contiguous()
        hp = h_prev.contiguous() if h_prev is not None else None
        h, u = torch.empty_like(f), torch.empty_like(f)
        StageUpdate._launch("tcfd_ns2d_stage_update", (f, hp, b), lin, coef, (h, u), f)
        ctx.lin, ctx.coef, ctx.has_prev = lin, coef, h_prev is not None
        return h, u

    @staticmethod
    def backward(ctx, g_h, g_u):
        if g_u is None:
            g_u = torch.zeros_like(g_h)
        if torch.is_grad_enabled():
            # create_graph=True: the stage is linear, its VJP as differentiable tensor ops (see FusedExplicitTerms.backward)
            fa, beta, gdt, mu, mud = ctx.coef
            lin = ctx.lin
            den = 1 / (1 - mud * lin)
            gh_total = gdt * den * g_u if g_h is None else g_h + gdt * den * g_u
            g_hp = beta * gh_total if (ctx.has_prev and ctx.needs_input_grad[1]) else None
            return fa * gh_total, g_hp, (1 + mu * lin) * den * g_u, None, None
        g_u = g_u.contiguous()
        gh = g_h.contiguous() if g_h is not None else None
        g_f, g_b = torch.empty_like(g_u), torch.empty_like(g_u)
        g_hp = torch.empty_like(g_u) if (ctx.has_prev and ctx.needs_input_grad[1]) else None
        StageUpdate._launch("tcfd_ns2d_stage_update_vjp", (g_u, gh), ctx.lin, ctx.coef, (g_f, g_hp, g_b), g_u)
        return g_f, g_hp, g_b, None, None


def _linear_term_on(op, plan, device):
    """The real (n, m) linear term in the plan's precision on the device, cached on the plan."""
    lin = op.linear_term
    key = (plan.rdtype, device, lin.data_ptr(), lin._version)
    cached = getattr(plan, "_lin_dev", None)
    if cached is None or cached[0] != key:
        cached = (key, lin.detach().to(device=device, dtype=plan.rdtype).contiguous())
        plan._lin_dev = cached
    return cached[1]


def fused_scheduled_steps(op, plan, w_hat, steps, sched, forcing_hat):
    """``scheduled_steps`` for a schedule of plain numbers (only the STATE requires grad) on fused nodes: the explicit
    terms (``FusedExplicitTerms``) and the stage update (``StageUpdate``), i.e. 4 launches forward and ~11 backward per stage."""
    n = len(sched["beta"])
    fa = sched.get("fa") or [1.0] * n
    mu_den = sched.get("mu_den") or sched["mu"]
    base0 = sched.get("base0") or [0] * n
    lin = _linear_term_on(op, plan, w_hat.device)
    u = w_hat
    for _ in range(steps):
        start, h = u, None
        for k in range(n):
            f = explicit_terms(op, plan, u, forcing_hat)
            b = start if base0[k] else u
            coef = (float(fa[k]), float(sched["beta"][k]), float(sched["gdt"][k]), float(sched["mu"][k]), float(mu_den[k]))
            h, u = StageUpdate.apply(f, h, b, lin, coef)
    return u


def scheduled_steps(op, plan, w_hat, steps, sched, forcing_hat):
    """The generic stage schedule (``IMEXStepper.stage_schedule``) with detached scalars -- gradients w.r.t. the state."""
    lin = op.linear_term.to(w_hat.device)
    n = len(sched["beta"])
    fa = sched.get("fa") or [1.0] * n
    mu_den = sched.get("mu_den") or sched["mu"]
    base0 = sched.get("base0") or [0] * n
    u = w_hat
    for _ in range(steps):
        start, h = u, None
        for k in range(n):
            f = explicit_terms(op, plan, u, forcing_hat)
            h = fa[k] * f if h is None else fa[k] * f + sched["beta"][k] * h
            b = start if base0[k] else u
            u = (b + sched["gdt"][k] * h + sched["mu"][k] * lin * b) / (1 - mu_den[k] * lin)
    return u
