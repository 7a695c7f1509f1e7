"""CPU-only: host-side logic of the operator API (tables, state_dict layout,
coefficient rounding, loud failure without a HIP device)."""
import math

import numpy as np
import pytest
import torch

import torch_cfd_amd as tc
from conftest import load_golden

L = 2 * math.pi


@pytest.fixture(autouse=True)
def _restore_default_dtype():
    old = torch.get_default_dtype()
    yield
    torch.set_default_dtype(old)


def make(n, real, forcing=False):
    torch.set_default_dtype(real)
    grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
    fn = tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=4) if forcing else None
    return grid, tc.NavierStokes2DSpectral(1e-3, grid, drag=0.1, forcing_fn=fn, solver=tc.RK4CrankNicolsonStepper())


@pytest.mark.parametrize("n", [8, 16, 64, 128])
@pytest.mark.parametrize("tag,real", [("f64", torch.float64), ("f32", torch.float32)])
def test_buffers_equal_reference_tables(n, tag, real):
    g = load_golden("ns2d_tables.npz")
    _, op = make(n, real)
    for key in ("kx", "ky", "laplace", "linear_term", "filter"):
        np.testing.assert_array_equal(getattr(op, key).numpy(), g[f"{key}_{n}_{tag}"], err_msg=key)


def test_state_dict_keys_match_reference():
    _, op = make(16, torch.float64)
    assert list(op.state_dict().keys()) == [
        "kx", "ky", "laplace", "linear_term", "filter",
        "solver.params.alphas", "solver.params.betas", "solver.params.gammas"]
    assert all(not p.requires_grad for p in op.solver.params.values())


def test_grid_domain_and_mesh():
    g = tc.Grid(shape=(8, 8), domain=((0, L), (0, L)))
    assert g.step == (L / 8, L / 8) and g.ndim == 2
    kx, ky = g.rfft_mesh()
    assert kx.shape == (8, 5) and ky[0, -1] < 0  # Nyquist column carries the negative frequency
    with pytest.raises(TypeError):
        tc.Grid(shape=(8, 8), step=1.0, domain=((0, 1), (0, 1)))
    with pytest.raises(ValueError):
        tc.Grid(shape=(8, 8), domain=((0, 1),))


def test_stage_scalars_round_like_the_reference():
    torch.set_default_dtype(torch.float32)
    s = tc.RK4CrankNicolsonStepper()
    beta, gdt, mu = tc.RK4CrankNicolsonStepper.stage_scalars(s.params, 1e-3)
    al, ga = s.params["alphas"], s.params["gammas"]
    assert gdt[1] == (ga[1] * 1e-3).item() and mu[2] == (0.5 * 1e-3 * (al[3] - al[2])).item()
    assert beta[0] == 0.0 and len(beta) == len(gdt) == len(mu) == 5
    assert float(np.float32(gdt[3])) == gdt[3]  # exactly representable in fp32


def test_classic_rk4_weights_constructible():
    # the reference raises here (integer betas, SURVEY bug 2); the float weights are the intended ones
    s = tc.RK4CrankNicolsonStepper(low_storage=False)
    assert s.params["betas"].dtype.is_floating_point and len(s.params["gammas"]) == 4


def test_forcing_table_matches_oracle():
    from oracle import ns2d as O

    for real in (torch.float64, torch.float32):
        _, op = make(32, real, forcing=True)
        t = O.make_tables(32, L, 1e-3, 0.1, True, None, real)
        ref = O.kolmogorov_forcing_hat(32, L, t.kx, t.ky, 1.0, 4, real=real)
        op.forcing_noise_floor = 0  # keep every entry: equals the reference's table
        fh = op.forcing_hat()
        assert fh.dtype == ref.dtype
        assert torch.allclose(fh, ref, rtol=0, atol=1e-12 if real == torch.float64 else 1e-4)
        op.forcing_noise_floor = None  # default: transform round-off (< n*eps*max) becomes exact zeros
        clean = op.forcing_hat()
        assert int((clean != 0).sum()) == 1  # sin(4y): one half-spectrum mode
        assert (clean - ref).abs().max() <= 32 * torch.finfo(real).eps * ref.abs().max()


def test_cpu_tensor_fails_loudly_no_fallback():
    _, op = make(16, torch.float64)
    w = torch.zeros(1, 16, 9, dtype=torch.complex128)
    with pytest.raises(tc._lib.TcfdError, match="HIP"):
        op(w, 1e-3)
    with pytest.raises(tc._lib.TcfdError):
        op.explicit_terms(w)


def test_stable_time_step():
    dx = L / 1024
    assert tc.stable_time_step(dx=dx, max_velocity=5.0) == pytest.approx(0.5 * dx / 5)
    assert tc.stable_time_step(dx=dx, dt=1e-3, max_velocity=0.1) == pytest.approx(1e-3)
    assert tc.stable_time_step(dx=0.1, implicit_diffusion=False, viscosity=1.0) == pytest.approx(0.1**2 / 4)


def test_product_code_never_imports_the_oracle():
    import os
    import re

    from conftest import ROOT

    pkg = os.path.join(ROOT, "torch-cfd_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f



def test_composite_transforms_for_grids_with_an_odd_factor():
    """mixed_radix.CompositeFft (n = p * 2^k through p^2 power-of-two transforms) with a torch.fft stand-in for the
    power-of-two plan: the decomposition itself, including torch's c2r semantics on spectra that are not Hermitian."""
    from torch_cfd_amd.mixed_radix import CompositeFft, odd_factor_split

    class Pow2:
        cdtype, rdtype = torch.complex128, torch.float64

        def rfft2(self, x):
            assert (x.shape[-1] & (x.shape[-1] - 1)) == 0
            return torch.fft.rfft2(x)

        def irfft2(self, xh):
            return torch.fft.irfft2(xh, s=(xh.shape[-2], xh.shape[-2]))

    assert odd_factor_split(1024) is None and odd_factor_split(20) is None and odd_factor_split(96) == (3, 32)
    g = torch.Generator().manual_seed(0)
    for n in (24, 40, 48, 56, 96):
        p, _ = odd_factor_split(n)
        f = CompositeFft(n, p, Pow2())
        y = torch.randn(3, n, n, generator=g, dtype=torch.float64)
        z = torch.complex(torch.randn(2, n, n // 2 + 1, generator=g, dtype=torch.float64),
                          torch.randn(2, n, n // 2 + 1, generator=g, dtype=torch.float64))
        assert (f.rfft2(y) - torch.fft.rfft2(y)).abs().max() < 1e-11
        assert (f.irfft2(z) - torch.fft.irfft2(z, s=(n, n))).abs().max() < 1e-13
