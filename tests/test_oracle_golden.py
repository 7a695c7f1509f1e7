"""Pin the CPU oracle (oracle/ns2d.py) against vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU-only; no HIP involved."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import ns2d as O

L = 2 * math.pi
REAL = {"f64": torch.float64, "f32": torch.float32}
# fp64: the oracle repeats the reference's op sequence, so agreement is at round-off.
TOL = {"f64": 1e-13, "f32": 2e-6}


def forcing_for(name, n, t, real):
    if name == "kolmogorov":
        return O.kolmogorov_forcing_hat(n, L, t.kx, t.ky, 1.0, 4, False, False, real=real)
    if name == "kolmogorov_vort":
        return O.kolmogorov_forcing_hat(n, L, t.kx, t.ky, 1.0, 2, False, True, real=real)
    if name == "sincos":
        return O.sincos_forcing_hat(n, L, 0.1, 1.0, diam=L, real=real)
    return None


def tables(n, real, forcing=None, drag=0.0):
    t = O.make_tables(n, L, 1e-3, drag, True, None, real)
    t.forcing_hat = forcing_for(forcing, n, t, real)
    return t


@pytest.mark.parametrize("n", [8, 16, 64, 128])
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_tables_match_reference(n, tag):
    g = load_golden("ns2d_tables.npz")
    t = O.make_tables(n, L, 1e-3, 0.1, True, None, REAL[tag])
    for key, val in (("kx", t.kx), ("ky", t.ky), ("laplace", t.laplace),
                     ("linear_term", t.linear_term), ("filter", t.mask)):
        ref = g[f"{key}_{n}_{tag}"]
        assert val.shape == ref.shape
        assert val.dtype == REAL[tag]
        np.testing.assert_array_equal(val.numpy(), ref, err_msg=key)


@pytest.mark.parametrize("n", [256, 512, 1024])
def test_mask_extents(n):
    g = load_golden("ns2d_tables.npz")
    mask = O.brick_wall_mask(n)
    np.testing.assert_array_equal(np.nonzero(mask[:, 0].numpy())[0], g[f"mask_rows_{n}"])
    np.testing.assert_array_equal(np.nonzero(mask[0, :].numpy())[0], g[f"mask_cols_{n}"])
    assert float(mask.sum()) == float(g[f"mask_sum_{n}"])


STEP_CASES = [(16, f, B) for f in (None, "kolmogorov", "sincos", "kolmogorov_vort") for B in (1, 3)] + \
             [(64, f, 2) for f in (None, "kolmogorov")]
DRAG = {None: 0.0, "kolmogorov": 0.1, "sincos": 0.0, "kolmogorov_vort": 0.05}


@pytest.mark.parametrize("n,forcing,B", STEP_CASES)
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_step_matches_reference(n, forcing, B, tag):
    g = load_golden("ns2d_steps.npz")
    key = f"n{n}_{tag}_{forcing}_B{B}"
    real = REAL[tag]
    t = tables(n, real, forcing, DRAG[forcing])
    w0 = torch.from_numpy(g[key + "_w0"])
    tol = TOL[tag]
    assert rel_l2(O.explicit_terms(w0, t), g[key + "_F"]) < tol
    w1, d1 = O.advance(w0, 1e-3, t)
    assert w1.dtype == w0.dtype
    assert rel_l2(w1, g[key + "_w1"]) < tol
    assert rel_l2(d1, g[key + "_dwdt1"]) < (1e-9 if tag == "f64" else 5e-3)
    w10, d10 = O.advance(w0, 1e-3, t, steps=10)
    assert rel_l2(w10, g[key + "_w10"]) < 10 * tol
    assert rel_l2(O.residual(w1, d1, t), g[key + "_res1"]) < (1e-7 if tag == "f64" else 5e-2)
    (uh, vh), psi = O.stream_and_velocity(w0, t.kx, t.ky)
    assert rel_l2(psi, g[key + "_psi"]) < tol
    if n == 16:
        assert rel_l2(uh, g[key + "_uh"]) < tol
        assert rel_l2(vh, g[key + "_vh"]) < tol


def test_4d_input():
    g = load_golden("ns2d_steps.npz")
    t = tables(16, torch.float64, "kolmogorov", 0.1)
    w1, d1 = O.advance(torch.from_numpy(g["n16_f64_4d_w0"]), 1e-3, t)
    assert w1.shape == (2, 3, 16, 9)
    assert rel_l2(w1, g["n16_f64_4d_w1"]) < 1e-13


def test_config1_kolmogorov128_200_steps():
    """BASELINE config 1 (reference CPU path): 128^2, B=1, fp64, forced, 200 steps."""
    g = load_golden("ns2d_c1_kolmogorov128.npz")
    t = tables(128, torch.float64, "kolmogorov", 0.1)
    w = torch.from_numpy(g["w0"])
    for step in range(1, 201):
        w, _ = O.advance(w, 1e-3, t)
        if step in (1, 10, 200):
            assert rel_l2(w, g[f"w{step}"]) < 1e-12, step


@pytest.mark.parametrize("n", [64, 128])
@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("seed", [0, 7])
def test_mcwilliams_ic(n, tag, seed):
    g = load_golden("ns2d_mcwilliams.npz")
    key = f"n{n}_{tag}_s{seed}"
    ic = O.mcwilliams_vorticity(n, L, 4, seed, REAL[tag])
    assert ic.dtype == REAL[tag]
    assert rel_l2(ic, g[key + "_ic"]) < (1e-13 if tag == "f64" else 1e-5)
    if key + "_w100" in g.files:
        t = tables(n, REAL[tag])
        w = torch.fft.rfft2(torch.from_numpy(g[key + "_ic"]))[None]
        w100, _ = O.advance(w, 1e-3, t, steps=100)
        assert rel_l2(w100, g[key + "_w100"]) < (1e-12 if tag == "f64" else 2e-4)


@pytest.mark.parametrize("n", [32, 64])
@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("seed,vmax,peak", [(0, 5.0, 4.0), (3, 1.0, 3.0)])
def test_filtered_velocity_ic(n, tag, seed, vmax, peak):
    """filtered_velocity_field + curl_2d (BASELINE config 1's initial condition) against the reference's output."""
    g = load_golden("ns2d_velocity_ic.npz")
    ux, uy = O.filtered_velocity_field(n, L, vmax, peak, 3, seed, REAL[tag])
    key = f"n{n}_{tag}_s{seed}"
    tol = 1e-13 if tag == "f64" else 1e-6
    assert rel_l2(ux, g[key + "_ux"]) < tol and rel_l2(uy, g[key + "_uy"]) < tol
    assert rel_l2(O.curl_2d(ux, uy, L), g[key + "_w"]) < tol
    # the projection leaves a field whose backward-difference divergence vanishes, at the prescribed max speed
    h = L / n
    div = (ux - torch.roll(ux, 1, 0)) / h + (uy - torch.roll(uy, 1, 1)) / h
    assert div.abs().max() < (1e-10 if tag == "f64" else 2e-3) * vmax / h
    assert abs(torch.sqrt(ux * ux + uy * uy).max().item() - vmax) < 1e-5 * vmax


def test_config1_initial_condition_is_the_velocity_ic():
    """The 128^2 Kolmogorov run's w0 fixture is rfft2(curl_2d(filtered_velocity_field(grid, 5, 4, seed 0)))."""
    g = load_golden("ns2d_c1_kolmogorov128.npz")
    ux, uy = O.filtered_velocity_field(128, L, 5.0, 4.0, 3, 0, torch.float64)
    assert rel_l2(torch.fft.rfft2(O.curl_2d(ux, uy, L))[None], g["w0"]) < 1e-13


@pytest.mark.parametrize("tag,cdt", [("f64", torch.complex128), ("f32", torch.complex64)])
def test_trajectory(tag, cdt):
    g = load_golden("ns2d_trajectory.npz")
    t = tables(32, REAL[tag], "kolmogorov", 0.1)
    out = O.trajectory(torch.from_numpy(g[f"{tag}_w0"]), 1e-3, t, num_steps=7, record_every_steps=3, dtype=cdt)
    for k in ("vorticity", "stream", "vort_t", "residual"):
        ref = g[f"{tag}_{k}"]
        assert tuple(out[k].shape) == ref.shape == (2, 3, 32, 17)
        assert out[k].dtype == cdt
        tol = {"vorticity": 1e-12, "stream": 1e-12, "vort_t": 1e-8, "residual": 1e-6} if tag == "f64" else \
              {"vorticity": 5e-6, "stream": 5e-6, "vort_t": 2e-2, "residual": 0.5}
        assert rel_l2(out[k], ref) < tol[k], k


def test_irfft2_ignores_imag_of_dc_and_nyquist_columns():
    """SURVEY note N2: c2r along the last axis drops Im of the DC / Nyquist bins
    AFTER the full complex inverse transform along axis -2."""
    g = load_golden("fft_semantics.npz")
    for n in (8, 16, 32):
        x = torch.from_numpy(g[f"x_{n}"])
        ref = torch.from_numpy(g[f"irfft2_{n}"])
        cols = torch.fft.ifft(x, dim=-2)
        cols[..., 0] = cols[..., 0].real + 0j
        cols[..., -1] = cols[..., -1].real + 0j
        herm = torch.cat([cols, cols[..., 1:-1].flip(-1).conj()], dim=-1)
        mine = torch.fft.ifft(herm, dim=-1)
        assert mine.imag.abs().max() < 1e-14
        assert rel_l2(mine.real, ref) < 1e-14
