#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Run only inside the build container (needs /root/reference, read-only):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The outputs (*.npz: inputs + the reference's outputs, no reference source) are
committed; nothing at test/bench time reads /root/reference.  Work-arounds for
the reference's latent bugs (SURVEY.md headline list): tqdm injected into
``solvers``, ``low_storage=True``, ``wave_number=`` instead of ``k=``, default
dtype set BEFORE constructing Grid/operator/IC.
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "fno", "data_gen"))
HERE = os.path.dirname(os.path.abspath(__file__))

from torch_cfd.grids import Grid  # noqa: E402
from torch_cfd.equations import NavierStokes2DSpectral, RK4CrankNicolsonStepper  # noqa: E402
from torch_cfd.forcings import KolmogorovForcing, SinCosForcing  # noqa: E402
from torch_cfd.initial_conditions import vorticity_field, filtered_velocity_field  # noqa: E402
from torch_cfd.finite_differences import curl_2d  # noqa: E402
from torch_cfd.spectral import vorticity_to_velocity  # noqa: E402

L = 2 * math.pi


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


def make_op(n, real, forcing=None, drag=0.0, nu=1e-3):
    torch.set_default_dtype(real)
    grid = Grid(shape=(n, n), domain=((0, L), (0, L)))
    if forcing == "kolmogorov":
        fn = KolmogorovForcing(grid=grid, scale=1.0, wave_number=4, swap_xy=False)
    elif forcing == "kolmogorov_vort":
        fn = KolmogorovForcing(grid=grid, scale=1.0, wave_number=2, swap_xy=False, vorticity=True)
    elif forcing == "sincos":
        fn = SinCosForcing(grid=grid, scale=0.1, k=1.0, diam=L)
    else:
        fn = None
    op = NavierStokes2DSpectral(viscosity=nu, grid=grid, drag=drag, smooth=True, forcing_fn=fn,
                                solver=RK4CrankNicolsonStepper())
    return grid, op


def ic_batch(grid, seeds, real):
    torch.set_default_dtype(real)
    w = torch.stack([vorticity_field(grid, 4, s).data for s in seeds])
    return torch.fft.rfft2(w)


def gen_tables():
    out = {}
    for n in (8, 16, 64, 128):
        for real, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            _, op = make_op(n, real, drag=0.1)
            for key in ("kx", "ky", "laplace", "linear_term", "filter"):
                out[f"{key}_{n}_{tag}"] = npy(getattr(op, key))
    for n in (256, 512, 1024):
        _, op = make_op(n, torch.float64)
        f = npy(op.filter)
        rows = np.nonzero(f[:, 0])[0]
        cols = np.nonzero(f[0, :])[0]
        out[f"mask_rows_{n}"] = rows.astype(np.int32)
        out[f"mask_cols_{n}"] = cols.astype(np.int32)
        out[f"mask_sum_{n}"] = np.array(f.sum())
    save("ns2d_tables.npz", **out)


def gen_steps():
    out = {}
    dt = 1e-3
    for n in (16, 64):
        for real, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            combos = ((None, 0.0), ("kolmogorov", 0.1), ("sincos", 0.0), ("kolmogorov_vort", 0.05))
            for forcing, drag in (combos if n == 16 else combos[:2]):
                for B in ((1, 3) if n == 16 else (2,)):
                    grid, op = make_op(n, real, forcing, drag)
                    w0 = ic_batch(grid, list(range(B)), real)
                    key = f"n{n}_{tag}_{forcing}_B{B}"
                    out[key + "_w0"] = npy(w0)
                    with torch.no_grad():
                        out[key + "_F"] = npy(op.explicit_terms(w0))
                        w1, d1 = op(w0, dt)
                        out[key + "_w1"] = npy(w1)
                        out[key + "_dwdt1"] = npy(d1)
                        w10, d10 = op(w0, dt, steps=10)
                        out[key + "_w10"] = npy(w10)
                        out[key + "_dwdt10"] = npy(d10)
                        out[key + "_res1"] = npy(op.residual(w1, d1))
                        (uh, vh), psi = vorticity_to_velocity(grid, w0, (op.kx, op.ky))
                        out[key + "_psi"] = npy(psi)
                        if n == 16:
                            out[key + "_uh"] = npy(uh)
                            out[key + "_vh"] = npy(vh)
    # 4-D input (B, T, n, m): the time axis marches in parallel (equations.py:454-457)
    grid, op = make_op(16, torch.float64, "kolmogorov", 0.1)
    w0 = ic_batch(grid, list(range(6)), torch.float64).reshape(2, 3, 16, 9)
    with torch.no_grad():
        w1, d1 = op(w0, dt)
    out["n16_f64_4d_w0"] = npy(w0)
    out["n16_f64_4d_w1"] = npy(w1)
    out["n16_f64_4d_dwdt1"] = npy(d1)
    save("ns2d_steps.npz", **out)


def gen_c1():
    """BASELINE config 1: Kolmogorov forced, 128^2, B=1, fp64, 200 steps."""
    torch.set_default_dtype(torch.float64)
    n = 128
    grid, op = make_op(n, torch.float64, "kolmogorov", 0.1)
    v0 = filtered_velocity_field(grid, 5, 4, random_state=0)
    w_phys = curl_2d(v0).data
    w0 = torch.fft.rfft2(w_phys)[None]
    out = {"w0": npy(w0), "forcing_hat": None}
    w = w0
    with torch.no_grad():
        for step in range(1, 201):
            w, _ = op(w, 1e-3)
            if step in (1, 10, 200):
                out[f"w{step}"] = npy(w)
    del out["forcing_hat"]
    save("ns2d_c1_kolmogorov128.npz", **out)


def gen_mcwilliams():
    out = {}
    for n in (64, 128):
        for real, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            for seed in (0, 7):
                torch.set_default_dtype(real)
                grid, op = make_op(n, real)
                w_phys = vorticity_field(grid, 4, seed).data
                key = f"n{n}_{tag}_s{seed}"
                out[key + "_ic"] = npy(w_phys)
                if n == 64 or (seed == 0 and real == torch.float64):
                    w = torch.fft.rfft2(w_phys)[None]
                    with torch.no_grad():
                        w1, _ = op(w, 1e-3)
                        out[key + "_w1"] = npy(w1)
                        w100, _ = op(w, 1e-3, steps=100)
                        out[key + "_w100"] = npy(w100)
    save("ns2d_mcwilliams.npz", **out)


def gen_velocity_ic():
    """filtered_velocity_field + curl_2d (initial_conditions.py:122-167, finite_differences.py:412-419): the staggered
    velocity components after the 3 project-and-normalise sweeps and the resulting vorticity."""
    out = {}
    for n in (32, 64):
        for real, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            for seed, vmax, peak in ((0, 5.0, 4.0), (3, 1.0, 3.0)):
                torch.set_default_dtype(real)
                grid = Grid(shape=(n, n), domain=((0, L), (0, L)))
                v = filtered_velocity_field(grid, vmax, peak, random_state=seed)
                key = f"n{n}_{tag}_s{seed}"
                out[key + "_ux"] = npy(v[0].data)
                out[key + "_uy"] = npy(v[1].data)
                out[key + "_w"] = npy(curl_2d(v).data)
    torch.set_default_dtype(torch.float64)
    save("ns2d_velocity_ic.npz", **out)


def gen_trajectory():
    import tqdm
    import solvers

    solvers.tqdm = tqdm.tqdm
    out = {}
    for real, tag, cdt in ((torch.float64, "f64", torch.complex128), (torch.float32, "f32", torch.complex64)):
        grid, op = make_op(32, real, "kolmogorov", 0.1)
        w0 = ic_batch(grid, [0, 1], real)
        with torch.no_grad():
            res = solvers.get_trajectory_imex(op, w0, 1e-3, num_steps=7, record_every_steps=3, dtype=cdt)
        out[f"{tag}_w0"] = npy(w0)
        for k, v in res.items():
            out[f"{tag}_{k}"] = npy(v)
    save("ns2d_trajectory.npz", **out)


def gen_irfft2():
    out = {}
    g = torch.Generator().manual_seed(123)
    for n in (8, 16, 32):
        m = n // 2 + 1
        x = torch.randn(2, n, m, 2, generator=g, dtype=torch.float64)
        xc = torch.view_as_complex(x)  # NOT Hermitian: DC/Nyquist bins carry imaginary parts
        out[f"x_{n}"] = npy(xc)
        out[f"irfft2_{n}"] = npy(torch.fft.irfft2(xc))
        r = torch.randn(2, n, n, generator=g, dtype=torch.float64)
        out[f"r_{n}"] = npy(r)
        out[f"rfft2_{n}"] = npy(torch.fft.rfft2(r))
    save("fft_semantics.npz", **out)


def gen_fno():
    """SpectralConv3d / SpectralConvS / SpectralConvT / SobolevLoss of the reference (fp32)."""
    torch.set_default_dtype(torch.float32)
    from fno.fno3d import SpectralConv3d
    from fno.sfno import SpectralConvS, SpectralConvT
    from fno.losses import SobolevLoss

    out = {}
    g = torch.Generator().manual_seed(7)

    def sd(mod, key):
        for k, v in mod.state_dict().items():
            out[f"{key}_sd_{k}"] = npy(v)

    with torch.no_grad():
        # (b, Ci, X, Y, T) = (2, 3, 16, 8, 10), Co = 4, modes (4, 3, 3): X != Y and Ci != Co catch transposes
        x = torch.randn(2, 3, 16, 8, 10, generator=g)
        torch.manual_seed(0)
        m = SpectralConv3d(3, 4, 4, 3, 3)
        out["conv3d_x"] = npy(x); sd(m, "conv3d"); out["conv3d_y"] = npy(m(x))

        for bias in (False, True):
            torch.manual_seed(1)
            m = SpectralConvS(3, 4, 4, 3, 3, bias=bias, delta=0.5)
            if bias:
                for b_ in m.bias:
                    b_.copy_(torch.randn(b_.shape, generator=g) * 0.05)
            key = f"convS_bias{int(bias)}"
            out[key + "_x"] = npy(x); sd(m, key); out[key + "_y"] = npy(m(x))

        for pad in (False, True):
            for steps in (10, 20, 40):
                torch.manual_seed(2)
                m = SpectralConvT(3, 4, 4, 3, 3, delta=0.1, bias=True, temporal_padding=pad)
                for b_ in m.bias:
                    b_.copy_(torch.randn(b_.shape, generator=g) * 0.05)
                key = f"convT_pad{int(pad)}_s{steps}"
                out[key + "_x"] = npy(x); sd(m, key); out[key + "_y"] = npy(m(x, out_steps=steps))

        # a BASELINE-config-5 shaped layer at reduced batch/width: (1, 4, 64, 64, 10), modes 24/24/5
        x5 = torch.randn(1, 4, 64, 64, 10, generator=g)
        torch.manual_seed(3)
        m = SpectralConvS(4, 4, 24, 24, 5)
        out["convS_c5_x"] = npy(x5); sd(m, "convS_c5"); out["convS_c5_y"] = npy(m(x5))

        a = torch.randn(2, 16, 16, 10, generator=g)
        b = torch.randn(2, 16, 16, 10, generator=g)
        out["sob_x"], out["sob_y"] = npy(a), npy(b)
        for order in (0, -1, 1):
            for rel in (True, False):
                out[f"sob_o{order}_r{int(rel)}"] = npy(SobolevLoss(n_grid=16, norm_order=order, relative=rel)(a, b))
    save("fno_layers.npz", **out)


def gen_sfno():
    """Tiny SFNO end to end (fp32): state_dict + input + outputs for two out_steps."""
    torch.set_default_dtype(torch.float32)
    from fno.sfno import SFNO

    out = {}
    torch.manual_seed(0)
    model = SFNO(4, 4, 3, width=4, num_spectral_layers=3, latent_steps=10).eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for b_ in model.output_operator.conv.bias:
            b_.copy_(torch.randn(b_.shape, generator=g) * 0.05)
        x = torch.randn(2, 16, 16, 10, generator=g)
        out["x"] = npy(x)
        for k, v in model.state_dict().items():
            out["sd_" + k] = npy(v)
        out["y10"] = npy(model(x))
        out["y20"] = npy(model(x, out_steps=20))
    save("fno_sfno_tiny.npz", **out)


def gen_sfno_padding():
    """Round 4: what the round-3 build raised on.  ``SFNO(spatial_padding=8)`` on 16^2 (the output convolution then runs on
    32^2) and on 24^2 (-> 40^2: no power of two anywhere), a 96^2 ``SpectralConvS`` layer, and the reference GRADIENTS of a
    spatially + temporally resampled ``SpectralConvS`` (fno/base.py:229-237 with ``out_mesh_size``)."""
    torch.set_default_dtype(torch.float32)
    from fno.sfno import SFNO, SpectralConvS

    out = {}
    g = torch.Generator().manual_seed(11)
    for n in (16, 24):
        torch.manual_seed(0)
        model = SFNO(4, 4, 3, width=4, num_spectral_layers=3, latent_steps=10, spatial_padding=8).eval()
        with torch.no_grad():
            for b_ in model.output_operator.conv.bias:
                b_.copy_(torch.randn(b_.shape, generator=g) * 0.05)
            x = torch.randn(2, n, n, 10, generator=g)
            out[f"pad{n}_x"] = npy(x)
            for k, v in model.state_dict().items():
                out[f"pad{n}_sd_" + k] = npy(v)
            out[f"pad{n}_y10"] = npy(model(x))
            out[f"pad{n}_y20"] = npy(model(x, out_steps=20))
    with torch.no_grad():
        x96 = torch.randn(1, 3, 96, 96, 10, generator=g)
        torch.manual_seed(4)
        m = SpectralConvS(3, 4, 12, 12, 5, bias=True, delta=0.5)
        for b_ in m.bias:
            b_.copy_(torch.randn(b_.shape, generator=g) * 0.05)
        out["c96_x"] = npy(x96)
        for k, v in m.state_dict().items():
            out["c96_sd_" + k] = npy(v)
        out["c96_y"] = npy(m(x96))
    # gradients of a resampled layer: (16, 16, 10) -> (24, 20, 12) and -> (12, 16, 8)
    for tag, size in (("up", [24, 20, 12]), ("down", [12, 16, 8])):
        torch.manual_seed(5)
        m = SpectralConvS(3, 4, 4, 3, 3)
        x = torch.randn(2, 3, 16, 16, 10, generator=g).requires_grad_(True)
        y = m(x, out_mesh_size=size)
        cot = torch.randn(y.shape, generator=g)
        (y * cot).sum().backward()
        out[f"rs_{tag}_x"], out[f"rs_{tag}_cot"], out[f"rs_{tag}_y"] = npy(x), npy(cot), npy(y)
        out[f"rs_{tag}_gx"] = npy(x.grad)
        for k, v in m.state_dict().items():
            out[f"rs_{tag}_sd_" + k] = npy(v)
        for k, prm in m.named_parameters():
            out[f"rs_{tag}_g_" + k] = npy(prm.grad)
    save("fno_sfno_padding.npz", **out)


def gen_grads():
    """Reference gradients (torch autograd through torch.fft on the CPU, fp32): SpectralConvS, SpectralConvT with
    temporal padding / resampling, and the tiny SFNO of gen_sfno under a SobolevLoss -- parameter and input grads."""
    torch.set_default_dtype(torch.float32)
    from fno.sfno import SFNO, SpectralConvS, SpectralConvT
    from fno.losses import SobolevLoss

    out = {}
    g = torch.Generator().manual_seed(11)
    for name, layer, xs, kw in (
        ("convS", SpectralConvS(3, 5, 4, 3, 3, bias=True, delta=0.3), (2, 3, 16, 8, 10), {}),
        ("convT_pad", SpectralConvT(4, 4, 4, 4, 3, delta=0.1, bias=True, temporal_padding=True), (2, 4, 16, 16, 6),
         {"out_steps": 9}),
        ("convT_plain", SpectralConvT(2, 3, 3, 4, 4, delta=0.1, bias=False, temporal_padding=False), (2, 2, 8, 16, 7),
         {"out_steps": 12}),
    ):
        with torch.no_grad():
            for p_ in layer.parameters():
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.2)
        x = torch.randn(*xs, generator=g).requires_grad_(True)
        y = layer(x, **kw)
        t = torch.randn(y.shape, generator=g)
        ((y * t).sum() + 0.5 * (y ** 2).sum()).backward()
        out[f"{name}_x"] = npy(x)
        out[f"{name}_t"] = npy(t)
        out[f"{name}_y"] = npy(y)
        out[f"{name}_gx"] = npy(x.grad)
        for k, v in layer.state_dict().items():
            out[f"{name}_sd_{k}"] = npy(v)
        for k, v in layer.named_parameters():
            out[f"{name}_g_{k}"] = npy(v.grad)
    # SpectralConvT with the Helmholtz projection between contraction and inverse transform (out_dim = 2 OutConv core)
    from fno.sfno import HelmholtzProjection
    hp = SpectralConvT(2, 2, 4, 4, 3, delta=0.1, bias=True, temporal_padding=True,
                       postprocess=HelmholtzProjection(n_grid=16, diam=2 * math.pi))
    with torch.no_grad():
        for p_ in hp.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.2)
    xh = torch.randn(2, 2, 16, 16, 6, generator=g).requires_grad_(True)
    yh = hp(xh, out_steps=9)
    th = torch.randn(yh.shape, generator=g)
    ((yh * th).sum() + 0.5 * (yh ** 2).sum()).backward()
    out["convT_helm_x"], out["convT_helm_t"], out["convT_helm_y"], out["convT_helm_gx"] = npy(xh), npy(th), npy(yh), npy(xh.grad)
    for k, v in hp.state_dict().items():
        out[f"convT_helm_sd_{k}"] = npy(v)
    for k, v in hp.named_parameters():
        out[f"convT_helm_g_{k}"] = npy(v.grad)
    torch.manual_seed(0)
    model = SFNO(4, 4, 3, width=4, num_spectral_layers=3, latent_steps=10).train()
    with torch.no_grad():
        for b_ in model.output_operator.conv.bias:
            b_.copy_(torch.randn(b_.shape, generator=g) * 0.05)
    x = torch.randn(2, 16, 16, 10, generator=g).requires_grad_(True)
    target = torch.randn(2, 16, 16, 10, generator=g)
    loss = SobolevLoss(n_grid=16, norm_order=0, relative=True)(model(x), target)
    loss.backward()
    out["sfno_x"], out["sfno_target"], out["sfno_loss"], out["sfno_gx"] = npy(x), npy(target), npy(loss), npy(x.grad)
    for k, v in model.state_dict().items():
        out["sfno_sd_" + k] = npy(v)
    for k, v in model.named_parameters():
        out["sfno_g_" + k] = npy(v.grad) if v.grad is not None else np.zeros(0, dtype=np.float32)
    save("fno_grads.npz", **out)


def gen_grads_wide():
    """Reference gradients at the widths the reference itself uses besides its default 10 -- 16 (fno/sfno_pytest.py:258-270)
    and 20 (examples/ex2_SFNO_train_fnodata.ipynb) -- with GELU (fno/train.py:303) and ReLU: tiny SFNOs under a SobolevLoss,
    loss + input gradient + every parameter gradient (torch autograd through torch.fft on the CPU, fp32)."""
    torch.set_default_dtype(torch.float32)
    from fno.sfno import SFNO
    from fno.losses import SobolevLoss

    out = {}
    for tag, width, act, layers in (("w16_gelu", 16, "GELU", 2), ("w16_relu", 16, "ReLU", 2), ("w20_gelu", 20, "GELU", 2)):
        g = torch.Generator().manual_seed(1600 + width + len(act))
        torch.manual_seed(width)
        model = SFNO(4, 4, 3, width=width, num_spectral_layers=layers, activation=act, latent_steps=10).train()
        with torch.no_grad():
            for b_ in model.output_operator.conv.bias:
                b_.copy_(torch.randn(b_.shape, generator=g) * 0.05)
        x = torch.randn(2, 16, 16, 10, generator=g).requires_grad_(True)
        target = torch.randn(2, 16, 16, 10, generator=g)
        pred = model(x)
        loss = SobolevLoss(n_grid=16, norm_order=0, relative=True)(pred, target)
        loss.backward()
        out[f"{tag}_x"], out[f"{tag}_target"], out[f"{tag}_pred"] = npy(x), npy(target), npy(pred)
        out[f"{tag}_loss"], out[f"{tag}_gx"] = npy(loss), npy(x.grad)
        for k, v in model.state_dict().items():
            out[f"{tag}_sd_" + k] = npy(v)
        for k, v in model.named_parameters():
            out[f"{tag}_g_" + k] = npy(v.grad) if v.grad is not None else np.zeros(0, dtype=np.float32)
    save("fno_grads_wide.npz", **out)


def gen_imex():
    """IMEXStepper orders 1 / 1.5 / 2 (equations.py:110-246) on the spectral operator, 3 steps, fp64."""
    from torch_cfd.equations import IMEXStepper

    torch.set_default_dtype(torch.float64)
    out = {}
    n = 32
    grid = Grid(shape=(n, n), domain=((0, L), (0, L)))
    w0 = ic_batch(grid, [0, 1], torch.float64)
    out["w0"] = npy(w0)
    for order, alpha, beta in ((1, 1.0, 1.0), (1.5, 0.5, 0.5), (2, 0.5, 0.5), (2, 2 / 3, 0.5)):
        fn = KolmogorovForcing(grid=grid, scale=1.0, wave_number=4, swap_xy=False)
        op = NavierStokes2DSpectral(viscosity=1e-3, grid=grid, drag=0.1, smooth=True, forcing_fn=fn,
                                    solver=IMEXStepper(order=order, alpha=alpha, beta=beta))
        with torch.no_grad():
            w, d = op(w0, 1e-3, steps=3)
        out[f"o{order}_a{alpha:.3f}_w"] = npy(w)
        out[f"o{order}_a{alpha:.3f}_dwdt"] = npy(d)
    save("ns2d_imex.npz", **out)


def gen_helmholtz():
    """SpectralConvT with the Helmholtz projection as spectrum post-processing (the out_dim = 2 OutConv core)."""
    torch.set_default_dtype(torch.float32)
    from fno.sfno import HelmholtzProjection, SpectralConvT

    out = {}
    g = torch.Generator().manual_seed(21)
    torch.manual_seed(4)
    n = 16
    m = SpectralConvT(2, 2, 4, 4, 3, delta=0.1, bias=True, temporal_padding=True,
                      postprocess=HelmholtzProjection(n_grid=n, diam=2 * math.pi))
    with torch.no_grad():
        for b_ in m.bias:
            b_.copy_(torch.randn(b_.shape, generator=g) * 0.05)
        x = torch.randn(2, 2, n, n, 6, generator=g)
        out["x"] = npy(x)
        for k, v in m.state_dict().items():
            out["sd_" + k] = npy(v)
        out["y"] = npy(m(x, out_steps=9))
    save("fno_helmholtz.npz", **out)


def gen_legacy_cn():
    """Legacy IMEX Crank-Nicolson step + residual (fno/data_gen/solvers.py:49-188), fp64."""
    import solvers

    torch.set_default_dtype(torch.float64)
    out = {}
    n = 32
    grid = Grid(shape=(n, n), domain=((0, L), (0, L)))
    w0 = ic_batch(grid, [0, 1, 2], torch.float64)
    fphys = SinCosForcing(grid=grid, scale=0.1, k=1.0, diam=L)(grid, None)
    fh = torch.fft.rfft2(fphys)
    out["w0"], out["f"] = npy(w0), npy(fh)
    for dealias in (False, True):
        w_next, dwdt, w, psi, res, (kx, ky), lap, filt = solvers.imex_crank_nicolson_step(
            w0, fh, 1e-3, 1e-3, diam=L, dealias=dealias, output_rfft=True)
        tag = f"d{int(dealias)}"
        for k, v in (("w_next", w_next), ("dwdt", dwdt), ("psi", psi), ("res", res)):
            out[f"{tag}_{k}"] = npy(v)
        res2 = solvers.update_residual(w_next, dwdt, fh, 1e-3, (kx, ky), lap, dealias_filter=filt, dealias=dealias)
        out[f"{tag}_res_update"] = npy(res2)
    save("ns2d_legacy_cn.npz", **out)


def gen_legacy_cn_trajectory():
    """The legacy driver around that step (fno/data_gen/solvers.py:268-448: record schedule, c2r, bilinear subsample) and
    ``backdiff`` (:19-35), fp64."""
    import solvers

    solvers.tqdm = __import__("tqdm").tqdm
    torch.set_default_dtype(torch.float64)
    out = {}
    n = 32
    grid = Grid(shape=(n, n), domain=((0, L), (0, L)))
    w0 = torch.stack([vorticity_field(grid, 4, s).data for s in (0, 1, 2)])
    f = SinCosForcing(grid=grid, scale=0.1, k=1.0, diam=L)(grid, None)
    out["w0"], out["f"] = npy(w0), npy(f)
    for tag, kw in (("s1", dict(subsample=1, dealias=True)), ("s2", dict(subsample=2, dealias=True)),
                    ("s1_nodealias", dict(subsample=1, dealias=False)), ("s2_one", dict(subsample=2, dealias=True))):
        ww = w0[:1] if tag == "s2_one" else w0
        res = solvers.get_trajectory_imex_crank_nicolson(ww, f, visc=1e-3, T=0.02, delta_t=1e-3, record_steps=4, diam=L,
                                                         pbar=False, **kw)
        for k, v in res.items():
            out[f"{tag}_{k}"] = npy(v)
    # a forcing per sample (B, n, n) and a time step that does not divide T (ceil(T / delta_t) = 13 steps, a record every 6)
    fb = torch.stack([f, 2.0 * f, -0.5 * f])
    res = solvers.get_trajectory_imex_crank_nicolson(w0, fb, visc=1e-3, T=0.0125, delta_t=1e-3, record_steps=2, diam=L, pbar=False,
                                                     subsample=2, dealias=True)
    for k, v in res.items():
        out[f"batched_f_{k}"] = npy(v)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 4, 4, 8, generator=g)
    out["bdf_x"] = npy(x)
    for order in (1, 2, 3, 4, 5):
        out[f"bdf_{order}"] = npy(solvers.backdiff(x, order))
    save("ns2d_legacy_cn_trajectory.npz", **out)


def gen_kolmogorov_dataset():
    """The batch loop of fno/data_gen/data_gen_Kolmogorov2d.py:119-192 at a small size, restated line by line from the
    reference's own components (the driver module itself needs h5py / xarray and is not importable here): Kolmogorov forcing,
    drag 0.1, per-sample filtered-velocity initial condition with the driver's seed rule, warm-up steps, get_trajectory_imex,
    irfft2 -> float32 -> bilinear subsample, random_states."""
    import solvers
    import torch.nn.functional as F

    solvers.tqdm = __import__("tqdm").tqdm
    torch.set_default_dtype(torch.float64)
    n, total_samples, batch_size, random_state, subsample = 32, 4, 2, 7, 2
    dt, warmup_steps, total_steps, record_every = 1e-3, 5, 6, 3
    ns = n // subsample
    grid = Grid(shape=(n, n), domain=((0, L), (0, L)))
    fn = KolmogorovForcing(grid=grid, scale=1.0, wave_number=4, swap_xy=False)
    ns2d = NavierStokes2DSpectral(viscosity=1e-3, grid=grid, drag=0.1, smooth=True, forcing_fn=fn, solver=RK4CrankNicolsonStepper())
    batches = []
    for i, idx in enumerate(range(0, total_samples, batch_size)):
        vort_init = torch.stack([curl_2d(filtered_velocity_field(grid, 5, 4, random_state=random_state + i + k)).data
                                 for k in range(batch_size)])
        vort_hat = torch.fft.rfft2(vort_init)
        for j in range(warmup_steps):
            vort_hat, _ = ns2d.step(vort_hat, dt)
        result = solvers.get_trajectory_imex(ns2d, vort_hat, dt, num_steps=total_steps, record_every_steps=record_every, pbar=False)
        for field, value in result.items():
            value = torch.fft.irfft2(value).real.cpu().to(torch.float32)
            result[field] = F.interpolate(value, size=(ns, ns), mode="bilinear")
        result["random_states"] = torch.tensor([random_state + idx + k for k in range(batch_size)], dtype=torch.int32)
        batches.append(result)
    out = {k: npy(torch.cat([b[k] for b in batches])) for k in batches[0]}
    out["params"] = np.array([n, total_samples, batch_size, random_state, subsample, warmup_steps, total_steps, record_every])
    save("ns2d_kolmogorov_dataset.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["tables", "steps", "c1", "mcwilliams", "velocity_ic", "trajectory", "irfft2", "fno", "sfno", "sfno_padding", "grads", "grads_wide", "imex",
                             "helmholtz", "legacy_cn", "legacy_cn_trajectory", "kolmogorov_dataset"]
    for w in which:
        globals()["gen_" + w]()
