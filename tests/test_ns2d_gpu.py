"""GPU parity tests of the fused RK4-CN spectral path (through the C ABI) against
(a) the committed reference-generated golden vectors, (b) the CPU oracle on the
same seeded inputs, (c) size-independent properties at BASELINE sizes.

Tolerances (rel-L2): fp64 <= 1e-10 per call (north_star bar: 1e-6); fp32 vs the
reference's own fp32 results <= 2e-6 @1 step, <= 1e-5 @10 steps, <= 5e-4 @1000
(SURVEY note N5).
"""
import math
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu

L = 2 * math.pi
REAL = {"f64": torch.float64, "f32": torch.float32}
CPLX = {"f64": torch.complex128, "f32": torch.complex64}
DRAG = {None: 0.0, "kolmogorov": 0.1, "sincos": 0.0, "kolmogorov_vort": 0.05}


@pytest.fixture(autouse=True)
def _restore_default_dtype():
    old = torch.get_default_dtype()
    yield
    torch.set_default_dtype(old)


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def scaled_err(a, b, scale_of):
    """||a - b|| / ||scale_of||: the error of a quantity that is a DIFFERENCE of larger terms, measured against the
    terms that cancel.  dw/dt = (w_new - w_old) / (k dt) inherits the round-off of w amplified by |w| / |w_new - w_old|
    (~1e2..1e3 at dt = 1e-3), and the PDE residual w_t - F - L w is a truncation-size remainder of terms of size
    |w_t|: a plain relative L2 of those in fp32 only measures that amplification (round 1 had to allow 5e-3 .. 0.5).
    Scaled by what cancels, fp32 dw/dt and residual must meet the SAME bound as w itself."""
    a, b, s = (torch.as_tensor(x).to("cpu", torch.complex128) for x in (a, b, scale_of))
    return (torch.linalg.norm((a - b).reshape(-1)) / torch.linalg.norm(s.reshape(-1))).item()


def build_op(n, tag, forcing, dev, drag=None, smooth=True, nu=1e-3):
    import torch_cfd_amd as tc

    torch.set_default_dtype(REAL[tag])
    grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
    fn = {None: None,
          "kolmogorov": lambda: tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=4, swap_xy=False),
          "kolmogorov_vort": lambda: tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=2, vorticity=True),
          "sincos": lambda: tc.SinCosForcing(grid=grid, scale=0.1, k=1.0, diam=L)}[forcing]
    fn = fn() if fn else None
    op = tc.NavierStokes2DSpectral(nu, grid, drag=DRAG[forcing] if drag is None else drag, smooth=smooth,
                                   forcing_fn=fn, solver=tc.RK4CrankNicolsonStepper()).to(dev)
    return grid, op


def oracle_tables(n, tag, forcing, drag=None, smooth=True, nu=1e-3):
    from oracle import ns2d as O

    real = REAL[tag]
    t = O.make_tables(n, L, nu, DRAG[forcing] if drag is None else drag, smooth, None, real)
    if forcing == "kolmogorov":
        t.forcing_hat = O.kolmogorov_forcing_hat(n, L, t.kx, t.ky, 1.0, 4, False, False, real=real)
    elif forcing == "kolmogorov_vort":
        t.forcing_hat = O.kolmogorov_forcing_hat(n, L, t.kx, t.ky, 1.0, 2, False, True, real=real)
    elif forcing == "sincos":
        t.forcing_hat = O.sincos_forcing_hat(n, L, 0.1, 1.0, diam=L, real=real)
    return t


# ----------------------------------------------------------------------------- golden vectors
STEP_CASES = [(16, f, B) for f in (None, "kolmogorov", "sincos", "kolmogorov_vort") for B in (1, 3)] + \
             [(64, f, 2) for f in (None, "kolmogorov")]


@pytest.mark.parametrize("n,forcing,B", STEP_CASES)
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_golden_steps(n, forcing, B, tag, dev):
    import torch_cfd_amd as tc

    g = load_golden("ns2d_steps.npz")
    key = f"n{n}_{tag}_{forcing}_B{B}"
    grid, op = build_op(n, tag, forcing, dev)
    w0 = torch.from_numpy(g[key + "_w0"]).to(dev)
    t1, t10 = (1e-10, 1e-10) if tag == "f64" else (2e-6, 1e-5)
    F = op.explicit_terms(w0)
    assert F.dtype == CPLX[tag] and F.shape == w0.shape
    assert rel_l2(F, g[key + "_F"]) < (1e-10 if tag == "f64" else 2e-6)
    w1, d1 = op(w0, 1e-3)
    assert w1.dtype == CPLX[tag] and w1.is_cuda
    assert rel_l2(w1, g[key + "_w1"]) < t1
    w1_ref = torch.from_numpy(g[key + "_w1"])
    if tag == "f64":
        assert rel_l2(d1, g[key + "_dwdt1"]) < 1e-8
    assert scaled_err(d1, g[key + "_dwdt1"], w1_ref / 1e-3) < t1          # same bound as w1 itself
    w10, d10 = op(w0, 1e-3, steps=10)
    assert rel_l2(w10, g[key + "_w10"]) < t10
    if tag == "f64":
        assert rel_l2(d10, g[key + "_dwdt10"]) < 1e-8
    assert scaled_err(d10, g[key + "_dwdt10"], w1_ref / 1e-2) < t10
    res1 = op.residual(w1, d1)
    if tag == "f64":
        assert rel_l2(res1, g[key + "_res1"]) < 1e-6
    assert scaled_err(res1, g[key + "_res1"], g[key + "_dwdt1"]) < (1e-10 if tag == "f64" else 3e-4)
    (uh, vh), psi = tc.vorticity_to_velocity(grid, w0, (op.kx, op.ky))
    assert rel_l2(psi, g[key + "_psi"]) < (1e-13 if tag == "f64" else 1e-6)
    if n == 16:
        assert rel_l2(uh, g[key + "_uh"]) < (1e-13 if tag == "f64" else 1e-6)
        assert rel_l2(vh, g[key + "_vh"]) < (1e-13 if tag == "f64" else 1e-6)
    # stepping one call at a time equals steps=10 in one call
    w = w0
    for _ in range(10):
        w, _ = op(w, 1e-3)
    assert rel_l2(w, w10) < (1e-14 if tag == "f64" else 1e-6)


def test_golden_4d_input(dev):
    g = load_golden("ns2d_steps.npz")
    _, op = build_op(16, "f64", "kolmogorov", dev)
    w1, d1 = op(torch.from_numpy(g["n16_f64_4d_w0"]).to(dev), 1e-3)
    assert w1.shape == (2, 3, 16, 9) and d1.shape == (2, 3, 16, 9)
    assert rel_l2(w1, g["n16_f64_4d_w1"]) < 1e-10
    assert rel_l2(d1, g["n16_f64_4d_dwdt1"]) < 1e-8


def test_unbatched_input(dev):
    g = load_golden("ns2d_steps.npz")
    _, op = build_op(16, "f64", "kolmogorov", dev)
    w0 = torch.from_numpy(g["n16_f64_kolmogorov_B1_w0"]).to(dev)[0]
    w1, d1 = op(w0, 1e-3)
    assert w1.shape == (16, 9)
    assert rel_l2(w1, g["n16_f64_kolmogorov_B1_w1"][0]) < 1e-10


def test_config1_kolmogorov128_200_steps(dev):
    """BASELINE configs[0] on the reference's own IC: 128^2, B=1, fp64, 200 steps."""
    g = load_golden("ns2d_c1_kolmogorov128.npz")
    _, op = build_op(128, "f64", "kolmogorov", dev)
    w = torch.from_numpy(g["w0"]).to(dev)
    w1, _ = op(w, 1e-3)
    assert rel_l2(w1, g["w1"]) < 1e-10
    w10, _ = op(w, 1e-3, steps=10)
    assert rel_l2(w10, g["w10"]) < 1e-10
    w200 = w
    for _ in range(200):
        w200, _ = op(w200, 1e-3)
    assert rel_l2(w200, g["w200"]) < 1e-9


@pytest.mark.parametrize("n", [64, 128])
@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("seed", [0, 7])
def test_mcwilliams_ic_and_100_steps(n, tag, seed, dev):
    import torch_cfd_amd as tc
    from torch_cfd_amd.initial_conditions import vorticity_field

    g = load_golden("ns2d_mcwilliams.npz")
    key = f"n{n}_{tag}_s{seed}"
    grid, op = build_op(n, tag, None, dev)
    ic = vorticity_field(grid, 4, seed, device=dev)
    assert ic.dtype == REAL[tag] and ic.shape == (n, n)
    assert rel_l2(ic, g[key + "_ic"]) < (1e-12 if tag == "f64" else 2e-4)  # two fp32 pipelines; k^2 amplifies high-k round-off
    if key + "_w100" in g.files:
        plan = tc.fft_plan(n, CPLX[tag], dev)
        w = plan.rfft2(torch.from_numpy(g[key + "_ic"]).to(dev))[None]
        w1, _ = op(w, 1e-3)
        assert rel_l2(w1, g[key + "_w1"]) < (1e-10 if tag == "f64" else 2e-6)
        w100, _ = op(w, 1e-3, steps=100)
        # fp32: reference-fp32 vs reference-fp64 is 8.7e-6 after 100 steps
        assert rel_l2(w100, g[key + "_w100"]) < (1e-9 if tag == "f64" else 5e-5)


@pytest.mark.parametrize("n", [32, 64])
@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("seed,vmax,peak", [(0, 5.0, 4.0), (3, 1.0, 3.0)])
def test_filtered_velocity_ic_on_device(n, tag, seed, vmax, peak, dev):
    """SURVEY 8f rank 1, second half: filtered_velocity_field + curl_2d generated on the device."""
    import torch_cfd_amd as tc
    from torch_cfd_amd.initial_conditions import filtered_velocity_field, curl_2d

    g = load_golden("ns2d_velocity_ic.npz")
    torch.set_default_dtype(REAL[tag])
    grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
    ux, uy = filtered_velocity_field(grid, vmax, peak, random_state=seed, device=dev)
    w = curl_2d((ux, uy), grid)
    # batched form: sample b uses seeds (s, s + 1) like a single call with random_state = s
    bx, by = filtered_velocity_field(grid, vmax, peak, batch_seeds=[seed, seed + 5], device=dev)
    key = f"n{n}_{tag}_s{seed}"
    tol = 1e-11 if tag == "f64" else 2e-4   # fp32: two different fp32 FFT pipelines through 3 projection sweeps
    assert ux.dtype == REAL[tag] and ux.shape == (n, n)
    assert rel_l2(ux, g[key + "_ux"]) < tol and rel_l2(uy, g[key + "_uy"]) < tol
    assert rel_l2(w, g[key + "_w"]) < tol * 10
    assert rel_l2(bx[0], g[key + "_ux"]) < tol and rel_l2(by[0], g[key + "_uy"]) < tol and bx.shape == (2, n, n)


def test_config1_from_device_initial_condition(dev):
    """BASELINE configs[0] end to end on the device: IC generator -> rfft2 -> 200 RK4-CN steps."""
    import torch_cfd_amd as tc
    from torch_cfd_amd.initial_conditions import filtered_velocity_field, curl_2d

    g = load_golden("ns2d_c1_kolmogorov128.npz")
    grid, op = build_op(128, "f64", "kolmogorov", dev)   # sets the default dtype (the forcing follows it, as in the reference)
    w_phys = curl_2d(filtered_velocity_field(grid, 5, 4, random_state=0, device=dev), grid)
    w0 = tc.fft_plan(128, torch.complex128, dev).rfft2(w_phys)[None]
    assert rel_l2(w0, g["w0"]) < 1e-11
    w200, _ = op(w0, 1e-3, steps=200)
    assert rel_l2(w200, g["w200"]) < 1e-8


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_trajectory_matches_reference(tag, dev):
    import torch_cfd_amd as tc

    g = load_golden("ns2d_trajectory.npz")
    _, op = build_op(32, tag, "kolmogorov", dev)
    out = tc.get_trajectory_imex(op, torch.from_numpy(g[f"{tag}_w0"]).to(dev), 1e-3, num_steps=7,
                                 record_every_steps=3, dtype=CPLX[tag])
    tol = {"vorticity": 1e-10, "stream": 1e-10, "vort_t": 1e-8, "residual": 1e-6} if tag == "f64" else \
          {"vorticity": 5e-6, "stream": 5e-6}
    for k in ("vorticity", "stream", "vort_t", "residual"):
        ref = g[f"{tag}_{k}"]
        assert tuple(out[k].shape) == ref.shape and out[k].dtype == CPLX[tag] and out[k].device.type == "cpu"
        if k in tol:
            assert rel_l2(out[k], ref) < tol[k], k
    # fp32 differences-of-large-terms against what cancels (see scaled_err): same 5e-6 as the vorticity itself for
    # dw/dt; the residual adds the round-off of F (2e-6 |F|, |F| ~ 1e2 |w_t| here)
    w_ref = torch.from_numpy(g[f"{tag}_vorticity"])
    assert scaled_err(out["vort_t"], g[f"{tag}_vort_t"], w_ref / 1e-3) < (1e-10 if tag == "f64" else 5e-6)
    assert scaled_err(out["residual"], g[f"{tag}_residual"], g[f"{tag}_vort_t"]) < (1e-9 if tag == "f64" else 5e-4)


def test_fft_semantics_golden(dev):
    import torch_cfd_amd as tc

    g = load_golden("fft_semantics.npz")
    for n in (8, 16, 32):
        plan = tc.fft_plan(n, torch.complex128, dev)
        assert rel_l2(plan.irfft2(torch.from_numpy(g[f"x_{n}"]).to(dev)), g[f"irfft2_{n}"]) < 1e-14
        assert rel_l2(plan.rfft2(torch.from_numpy(g[f"r_{n}"]).to(dev)), g[f"rfft2_{n}"]) < 1e-14


# ----------------------------------------------------------------------------- oracle, larger sizes
@pytest.mark.parametrize("n,B", [(8, 5), (32, 4), (256, 4), (512, 2), (1024, 2), (2048, 1)])
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_against_oracle(n, B, tag, dev):
    from oracle import ns2d as O

    forcing = "kolmogorov" if n >= 16 else None
    _, op = build_op(n, tag, forcing, dev)
    t = oracle_tables(n, tag, forcing)
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 100 + s, REAL[tag])) for s in range(B)])
    steps = 3 if n <= 512 else 1
    ref, ref_dt = O.advance(w0, 1e-3, t, steps=steps)
    out, out_dt = op(w0.to(dev), 1e-3, steps=steps)
    assert rel_l2(out, ref) < (1e-10 if tag == "f64" else 4e-6)
    if tag == "f64":
        assert rel_l2(out_dt, ref_dt) < 1e-8
    assert scaled_err(out_dt, ref_dt, ref / (steps * 1e-3)) < (1e-10 if tag == "f64" else 4e-6)   # bound of `out` itself
    # fp32: the reference's forcing table carries ~1e-6 |F| of transform round-off in every bin, which the
    # operator zeroes (NavierStokes2DSpectral.forcing_noise_floor)
    assert rel_l2(op.explicit_terms(w0.to(dev)), O.explicit_terms(w0, t)) < (1e-10 if tag == "f64" else 2e-5)


def test_smooth_false_and_no_drag(dev):
    from oracle import ns2d as O

    _, op = build_op(64, "f64", None, dev, drag=0.0, smooth=False)
    t = oracle_tables(64, "f64", None, drag=0.0, smooth=False)
    w0 = torch.fft.rfft2(O.mcwilliams_vorticity(64, L, 4, 3, torch.float64))[None]
    out, _ = op(w0.to(dev), 1e-3, steps=2)
    ref, _ = O.advance(w0, 1e-3, t, steps=2)
    assert rel_l2(out, ref) < 1e-10
    # drag = 0: the mean mode is conserved exactly
    assert out[0, 0, 0].item() == pytest.approx(w0[0, 0, 0].item(), abs=1e-12)


def test_config2_mcwilliams256_b16_fp32_long_run(dev):
    """BASELINE configs[1] shape (256^2, B=16, fp32): 200 steps against the fp32 oracle."""
    from oracle import ns2d as O

    n, B = 256, 16
    _, op = build_op(n, "f32", None, dev)
    t = oracle_tables(n, "f32", None)
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, torch.float64).float()) for s in range(B)])
    ref, _ = O.advance(w0, 1e-3, t, steps=200)
    out, _ = op(w0.to(dev), 1e-3, steps=200)
    # two fp32 evaluations of a chaotic flow: the reference's own fp32-vs-fp64 drift is 2e-5 here
    assert rel_l2(out, ref) < 1e-4


# ----------------------------------------------------------------------------- properties at full size
def test_taylor_green_analytic_decay(dev):
    """omega0 = 2k cos(kx) cos(ky) is an exact solution decaying like exp(-2 nu k^2 t);
    the oracle reproduces it to 2.9e-10."""
    import torch_cfd_amd as tc

    n, k, nu, dt, steps = 64, 2, 1e-2, 1e-2, 100
    _, op = build_op(n, "f64", None, dev, drag=0.0, nu=nu)
    x = torch.arange(n, dtype=torch.float64) * (L / n)
    X, Y = torch.meshgrid(x, x, indexing="ij")
    w0 = 2 * k * torch.cos(k * X) * torch.cos(k * Y)
    plan = tc.fft_plan(n, torch.complex128, dev)
    wh = plan.rfft2(w0.to(dev))[None]
    # F vanishes identically for a single Fourier mode pair
    assert torch.linalg.norm(op.explicit_terms(wh)).item() < 1e-9 * torch.linalg.norm(wh).item()
    out, _ = op(wh, dt, steps=steps)
    exact = w0 * math.exp(-2 * nu * k * k * dt * steps)
    assert rel_l2(plan.irfft2(out)[0], exact) < 1e-9


def test_config3_full_size_batch_consistency(dev):
    """BASELINE configs[2] size (1024^2, B=64, fp64): every field of the batch must
    evolve exactly as it does alone (independent trajectories), the first two are
    checked against the oracle, the mask is idempotent and the result finite."""
    from oracle import ns2d as O
    import torch_cfd_amd as tc
    from torch_cfd_amd.initial_conditions import vorticity_field

    n, B = 1024, 64
    grid, op = build_op(n, "f64", "kolmogorov", dev)
    plan = tc.fft_plan(n, torch.complex128, dev)
    w0 = torch.cat([plan.rfft2(vorticity_field(grid, 4, batch_seeds=list(range(i, i + 8)), device=dev))
                    for i in range(0, B, 8)])
    dt = 6.136e-4
    out, dwdt = op(w0, dt)
    assert torch.isfinite(torch.view_as_real(out)).all()
    for idx in (0, 37, 63):
        alone, _ = op(w0[idx:idx + 1].clone(), dt)
        assert torch.equal(alone[0], out[idx])
    t = oracle_tables(n, "f64", "kolmogorov")
    ref, ref_dt = O.advance(w0[:2].cpu(), dt, t)
    assert rel_l2(out[:2], ref) < 1e-10
    assert rel_l2(dwdt[:2], ref_dt) < 1e-8
    # F is exactly band-limited: masking it again changes nothing
    F = op.explicit_terms(w0[:4]) - op.forcing_hat().to(dev)
    assert torch.equal(F * op.filter, F)


@pytest.mark.parametrize("tag,tol", [("f64", 1e-10), ("f32", 1e-5)])
def test_1024_ten_steps_in_one_call_against_oracle(tag, tol, dev):
    """1024^2, B = 2, TEN steps through forward(w, dt, steps=10) (the chunked multi-step path of the headline size)
    against the oracle: fp64 1e-10, fp32 the stated ten-step tolerance 1e-5."""
    from oracle import ns2d as O

    n, dt = 1024, 6.136e-4
    real = REAL[tag]
    grid, op = build_op(n, tag, "kolmogorov", dev)
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, real)) for s in range(2)])
    out, dwdt = op(w0.to(dev), dt, steps=10)
    t = oracle_tables(n, tag, "kolmogorov")
    ref, ref_dt = O.advance(w0, dt, t, steps=10)
    assert rel_l2(out, ref) < tol
    assert scaled_err(dwdt, ref_dt, ref / (10 * dt)) < tol


def test_config3_steps_in_one_call_equal_per_call_steps(dev):
    """BASELINE configs[2] size (1024^2, B = 64, fp64): forward(w, dt, steps=3) -- every chunk runs its three steps back
    to back, the state stays on die -- must equal three forward(w, dt) calls bit for bit, dw/dt to round-off."""
    import torch_cfd_amd as tc
    from torch_cfd_amd.initial_conditions import vorticity_field

    n, B, dt = 1024, 64, 6.136e-4
    grid, op = build_op(n, "f64", "kolmogorov", dev)
    plan = tc.fft_plan(n, torch.complex128, dev)
    w0 = torch.cat([plan.rfft2(vorticity_field(grid, 4, batch_seeds=list(range(i, i + 8)), device=dev))
                    for i in range(0, B, 8)])
    fused, dw_fused = op(w0, dt, steps=3)
    w = w0
    for _ in range(3):
        w, _ = op(w, dt)
    assert torch.equal(fused, w)
    assert rel_l2(dw_fused, (w - w0) / (3 * dt)) < 1e-12


def test_config4_shard_full_size(dev):
    """BASELINE configs[3], one GPU's shard: McWilliams decaying turbulence, 512^2, 64 fields, fp64, unforced, dt = 1e-3,
    ICs generated on the device (seeds 0..63).  12 steps through the trajectory API (fused steps between records):
    the first two fields against the oracle, every record field present, each field evolves as it does alone."""
    from oracle import ns2d as O
    import torch_cfd_amd as tc
    from torch_cfd_amd.initial_conditions import vorticity_field

    n, B, dt = 512, 64, 1e-3
    grid, op = build_op(n, "f64", None, dev, drag=0.0)
    plan = tc.fft_plan(n, torch.complex128, dev)
    w0 = torch.cat([plan.rfft2(vorticity_field(grid, 4, batch_seeds=list(range(i, i + 16)), device=dev))
                    for i in range(0, B, 16)])
    ic_ref = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, torch.float64)) for s in range(2)])
    assert rel_l2(w0[:2], ic_ref) < 1e-11
    traj = tc.get_trajectory_imex(op, w0, dt, num_steps=12, record_every_steps=5, dtype=torch.complex128, to_cpu=False)
    assert {k: tuple(v.shape) for k, v in traj.items()} == {k: (B, 3, n, n // 2 + 1) for k in
                                                           ("vorticity", "stream", "vort_t", "residual")}
    t = oracle_tables(n, "f64", None, drag=0.0)
    ref = O.trajectory(ic_ref, dt, t, num_steps=12, record_every_steps=5, dtype=torch.complex128)
    for k, tol in (("vorticity", 1e-10), ("stream", 1e-10), ("vort_t", 1e-8), ("residual", 1e-6)):
        assert rel_l2(traj[k][:2], ref[k]) < tol, k
    w12, _ = op(w0, dt, steps=11)   # records are taken after steps 1, 6, 11
    assert rel_l2(w12, traj["vorticity"][:, 2]) < 1e-13
    for idx in (0, 41, 63):
        alone, _ = op(w0[idx:idx + 1].clone(), dt, steps=11)
        assert torch.equal(alone[0], w12[idx])


def test_config4_dataset_generation_512(dev, tmp_path):
    """The data-generation loop of BASELINE configs[3] at its real grid size with a short schedule: 512^2, 8 samples
    in batches of 4, 3 warm-up + 11 recorded steps (records every 5), 4x subsample, float32 / complex64 storage --
    against the oracle running the same loop on the CPU for the first batch."""
    from oracle import ns2d as O
    from torch_cfd_amd.data_gen import generate_mcwilliams_dataset

    torch.set_default_dtype(torch.float64)
    n, dt = 512, 1e-3
    path = str(tmp_path / "c4.pt")
    data = generate_mcwilliams_dataset(n, total_samples=8, batch_size=4, dt=dt, warmup_steps=3, total_steps=11,
                                       record_every_steps=5, subsample=4, device=dev, path=path)
    assert sorted(data) == ["random_states", "residual", "stream", "vort_t", "vorticity"]
    assert data["vorticity"].shape == (8, 3, 128, 128) and data["vorticity"].dtype == torch.float32
    assert data["random_states"].tolist() == list(range(8))
    saved = torch.load(path)
    assert all(torch.equal(saved[k], data[k]) for k in data)
    t = O.make_tables(n, L, 1e-3, 0.0, True, None, torch.float64)
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, torch.float64)) for s in range(4)])
    w0, _ = O.advance(w0, dt, t, steps=3)
    ref = O.trajectory(w0, dt, t, num_steps=11, record_every_steps=5, dtype=torch.complex64)
    phys = {k: torch.nn.functional.interpolate(torch.fft.irfft2(v).float(), size=(128, 128), mode="bilinear")
            for k, v in ref.items()}
    for k in ("vorticity", "stream"):
        assert rel_l2(data[k][:4], phys[k]) < 1e-6, k
    # stored through complex64 records: dw/dt carries the storage round-off of w amplified by 1 / dt
    assert scaled_err(data["vort_t"][:4], phys["vort_t"], phys["vorticity"] / dt) < 1e-6
    assert scaled_err(data["residual"][:4], phys["residual"], phys["vort_t"]) < 1e-3


@pytest.mark.parametrize("n,factor", [(64, 2), (256, 2), (256, 4), (512, 2), (512, 4), (512, 8), (1024, 2)])
@pytest.mark.parametrize("cdtype", [torch.complex64, torch.complex128])
def test_fused_c2r_subsample_is_bit_identical_to_the_two_steps(dev, n, factor, cdtype):
    """Output side of the data-generation drivers (fno/data_gen/data_gen_McWilliams2d.py:158-163): irfft2 and the bilinear
    subsample as ONE pass (tcfd_irfft2_subsample) against the HIP irfft2 followed by F.interpolate -- equal bit for bit at factor 2, to rounding beyond --
    and against torch.fft + F.interpolate on the CPU in the same precision."""
    import torch_cfd_amd as tc
    from torch_cfd_amd.data_gen import spectral_to_physical
    from torch_cfd_amd.equations import fft_plan

    g = torch.Generator().manual_seed(n + factor)
    rdtype = torch.float32 if cdtype == torch.complex64 else torch.float64
    x = torch.randn(3, 2, n, n, generator=g, dtype=rdtype)
    xh = torch.fft.rfft2(x).to(dev)
    plan = fft_plan(n, cdtype, dev)
    assert plan.subsample_factor(n // factor) == factor
    assert plan.subsample_factor(n // 2 - 8) == 0 and plan.subsample_factor(n) == 0 and plan.subsample_factor(None) == 0
    fused = plan.irfft2_subsample(xh, factor)
    two = torch.nn.functional.interpolate(plan.irfft2(xh).reshape(-1, 1, n, n), size=(n // factor, n // factor),
                                          mode="bilinear").reshape(3, 2, n // factor, n // factor)
    assert fused.shape == two.shape and fused.dtype == rdtype
    if factor == 2:
        assert torch.equal(fused, two)       # same row pairs share a complex transform in both paths, same arithmetic after it
    else:                                    # rows S r + S/2 - 1 and S r + S/2 ride through ONE transform here, through two
        assert rel_l2(fused, two) < (3e-7 if rdtype == torch.float32 else 5e-16)     # different ones there: rounding only
    assert torch.equal(spectral_to_physical(xh, n // factor, rdtype), fused)        # the driver-side helper takes the fused pass
    cpu = torch.nn.functional.interpolate(x.reshape(-1, 1, n, n), size=(n // factor, n // factor), mode="bilinear")
    tol = 2e-6 if rdtype == torch.float32 else 1e-14
    assert rel_l2(fused.cpu().reshape(cpu.shape), cpu) < tol
    with pytest.raises(tc._lib.TcfdError):
        plan.irfft2_subsample(xh, 3)


@pytest.mark.parametrize("n,factor", [(256, 32), (160, 16), (80, 8), (64, 16), (128, 32)])
def test_subsample_factors_beyond_the_fused_kernel_take_the_two_step_path(dev, n, factor):
    """The one-pass c2r + subsample takes a factor up to the lanes of one row transform (16 at n = 256, 8 at n = 64 / 160,
    4 at n = 80: tcfd_irfft2_subsample_max_factor); larger power-of-two factors ran before that kernel existed and must
    keep running -- irfft2 + F.interpolate -- instead of raising (fno/data_gen/data_gen_McWilliams2d.py:158-163 takes any size)."""
    import torch_cfd_amd as tc
    from torch_cfd_amd.data_gen import spectral_to_physical
    from torch_cfd_amd.equations import fft_plan

    g = torch.Generator().manual_seed(n + factor)
    x = torch.randn(2, n, n, generator=g, dtype=torch.float64)
    xh = torch.fft.rfft2(x).to(dev)
    plan = fft_plan(n, torch.complex128, dev)
    limit = plan.lib.tcfd_irfft2_subsample_max_factor(plan.handle)
    assert 2 <= limit < factor and plan.subsample_factor(n // factor) == 0 and plan.subsample_factor(n // limit) == limit
    with pytest.raises(tc._lib.TcfdError):
        plan.irfft2_subsample(xh, factor)
    got = spectral_to_physical(xh, n // factor, torch.float64)
    ref = torch.nn.functional.interpolate(x.reshape(-1, 1, n, n), size=(n // factor, n // factor), mode="bilinear").reshape(2, n // factor, n // factor)
    assert got.shape == ref.shape and rel_l2(got.cpu(), ref) < 1e-13


def test_linearity_of_transforms_and_roundtrip(dev):
    import torch_cfd_amd as tc

    for n, cdt in ((1024, torch.complex128), (512, torch.complex64)):
        plan = tc.fft_plan(n, cdt, dev)
        g = torch.Generator(device="cpu").manual_seed(n)
        real = torch.float64 if cdt == torch.complex128 else torch.float32
        a = torch.randn(3, n, n, generator=g, dtype=real).to(dev)
        b = torch.randn(3, n, n, generator=g, dtype=real).to(dev)
        tol = 1e-13 if real == torch.float64 else 2e-6
        assert rel_l2(plan.rfft2(2 * a - 3 * b), 2 * plan.rfft2(a) - 3 * plan.rfft2(b)) < tol
        assert rel_l2(plan.irfft2(plan.rfft2(a)), a) < tol
        # Parseval with Hermitian weights
        ah = plan.rfft2(a)
        wts = torch.full((n // 2 + 1,), 2.0, dtype=real, device=dev)
        wts[0] = wts[-1] = 1.0
        lhs = (ah.abs() ** 2 * wts).sum().item() / (n * n)
        assert lhs == pytest.approx((a.double() ** 2).sum().item(), rel=1e-12 if real == torch.float64 else 1e-5)


def test_errors_are_loud(dev):
    import torch_cfd_amd as tc

    _, op = build_op(16, "f64", None, dev)
    with pytest.raises(ValueError):
        op(torch.zeros(2, 16, 8, dtype=torch.complex128, device=dev), 1e-3)  # wrong m
    with pytest.raises(tc._lib.TcfdError):          # no CPU fallback
        op(torch.zeros(1, 16, 9, dtype=torch.complex128), 1e-3)
    w = torch.zeros(1, 16, 9, dtype=torch.complex128, device=dev, requires_grad=True)
    assert op(w, 1e-3)[0].requires_grad            # round 1 raised here; gradients now go through autograd.py
    torch.set_default_dtype(torch.float64)
    grid = tc.Grid(shape=(20, 20), domain=((0, L), (0, L)))   # 5 * 4: the power-of-two part is below 8 -> dense transforms
    small = tc.NavierStokes2DSpectral(1e-3, grid, solver=tc.RK4CrankNicolsonStepper()).to(dev)
    out, _ = small(torch.zeros(1, 20, 11, dtype=torch.complex128, device=dev), 1e-3)
    assert out.shape == (1, 20, 11) and float(out.abs().max()) == 0.0
    grid = tc.Grid(shape=(4098, 4098), domain=((0, L), (0, L)))   # beyond every transform this library has
    bad = tc.NavierStokes2DSpectral(1e-3, grid, solver=tc.RK4CrankNicolsonStepper()).to(dev)
    with pytest.raises(tc._lib.TcfdError, match="n = 4098"):
        bad(torch.zeros(1, 4098, 2050, dtype=torch.complex128, device=dev), 1e-3)


class _DenseVorticityForcing(torch.nn.Module):
    """A state-independent forcing with energy at EVERY wavenumber (outside the 2/3 mask too)."""
    vorticity = True

    def __init__(self, n, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.field = torch.randn(n, n, generator=g, dtype=torch.float64) * 0.05

    def forward(self, grid, field=None):
        from torch_cfd_amd.forcings import FieldArray

        return FieldArray(self.field, (0, 0), grid)


@pytest.mark.parametrize("n", [64, 256])
def test_general_tables_dense_forcing_and_non_separable_linear_term(n, dev):
    """The kernels take compact forms (separable mask / linear term, sparse forcing, mask pruning) only
    when the tables allow it exactly; arbitrary tables go through the general path."""
    import torch_cfd_amd as tc
    from oracle import ns2d as O

    torch.set_default_dtype(torch.float64)
    grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
    fn = _DenseVorticityForcing(n, 11)
    op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.05, forcing_fn=fn, solver=tc.RK4CrankNicolsonStepper())
    with torch.no_grad():
        op.linear_term -= 1e-4 * (op.kx * op.ky) ** 2  # cross term: not of the form a[i] + b[j]
    op = op.to(dev)
    t = O.make_tables(n, L, 1e-3, 0.05, True, torch.fft.rfft2(fn.field), torch.float64)
    t.linear_term = t.linear_term - 1e-4 * (t.kx * t.ky) ** 2
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, torch.float64)) for s in range(3)])
    ref, ref_dt = O.advance(w0, 1e-3, t, steps=3)
    out, out_dt = op(w0.to(dev), 1e-3, steps=3)
    assert rel_l2(out, ref) < 1e-10
    assert rel_l2(op.explicit_terms(w0.to(dev)), O.explicit_terms(w0, t)) < 1e-10
    assert rel_l2(op.residual(out, out_dt), O.residual(ref, ref_dt, t)) < 1e-6


def test_compact_and_general_paths_agree(dev, monkeypatch):
    """Same operator through the pruned/separable kernels and through the general tables."""
    from oracle import ns2d as O

    n = 128
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, torch.float64)) for s in range(2)]).to(dev)
    _, op = build_op(n, "f64", "kolmogorov", dev)
    fast, _ = op(w0, 1e-3, steps=4)
    for var in ("TCFD_FORCE_TABLES", "TCFD_NO_PRUNE"):
        monkeypatch.setenv(var, "1")
        _, op2 = build_op(n, "f64", "kolmogorov", dev)
        slow, _ = op2(w0, 1e-3, steps=4)
        monkeypatch.delenv(var)
        assert rel_l2(slow, fast) < 1e-13, var


def test_trajectory_fuses_steps_between_records(dev):
    """record_every_steps > 1: the unrecorded steps run as one multi-step call; results equal the oracle's
    step-by-step loop, and a 1000-step fp32 run (BASELINE configs[1] length) stays within the stated drift."""
    import torch_cfd_amd as tc
    from oracle import ns2d as O

    n, B = 64, 3
    _, op = build_op(n, "f64", "kolmogorov", dev)
    t = oracle_tables(n, "f64", "kolmogorov")
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, torch.float64)) for s in range(B)])
    ref = O.trajectory(w0, 1e-3, t, num_steps=23, record_every_steps=5, dtype=torch.complex128)
    out = tc.get_trajectory_imex(op, w0.to(dev), 1e-3, num_steps=23, record_every_steps=5, dtype=torch.complex128)
    for k in ("vorticity", "stream", "vort_t", "residual"):
        assert out[k].shape == ref[k].shape == (B, 5, n, n // 2 + 1)
        assert rel_l2(out[k], ref[k]) < (1e-10 if k in ("vorticity", "stream") else 1e-6), k


def test_config2_1000_steps_fp32(dev):
    """BASELINE configs[1]: McWilliams 256^2, B=16, fp32, 1000 RK4-CN steps; stated tolerance 5e-4 vs the
    reference-equivalent fp32 run."""
    from oracle import ns2d as O

    n, B = 256, 16
    _, op = build_op(n, "f32", None, dev)
    t = oracle_tables(n, "f32", None)
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, torch.float64).float()) for s in range(B)])
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref, _ = O.advance(w0, 1e-3, t, steps=1000)
    out, _ = op(w0.to(dev), 1e-3, steps=1000)
    assert rel_l2(out, ref) < 5e-4


@pytest.mark.parametrize("order,alpha,beta", [(1, 1.0, 1.0), (1.5, 0.5, 0.5), (2, 0.5, 0.5), (2, 2 / 3, 0.5)])
def test_imex_steppers_golden(order, alpha, beta, dev, monkeypatch):
    """IMEXStepper orders 1 / 1.5 / 2 (equations.py:174-228) behind the same operator, every stage
    fused in the HIP step kernels through the general schedule h <- fa F + beta h, u <- (base + gdt h + mu L base) /
    (1 - mu_den L) with base = current or step-initial state."""
    import torch_cfd_amd as tc

    g = load_golden("ns2d_imex.npz")
    torch.set_default_dtype(torch.float64)
    grid = tc.Grid(shape=(32, 32), domain=((0, L), (0, L)))
    fn = tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=4)
    op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.1, forcing_fn=fn,
                                   solver=tc.IMEXStepper(order=order, alpha=alpha, beta=beta)).to(dev)
    w0 = torch.from_numpy(g["w0"]).to(dev)
    w, d = op(w0, 1e-3, steps=3)                       # fused schedule (tcfd_ns2d_step_imex), graph replay inside
    assert rel_l2(w, g[f"o{order}_a{alpha:.3f}_w"]) < 1e-10
    assert rel_l2(d, g[f"o{order}_a{alpha:.3f}_dwdt"]) < 1e-8
    # the fused step against the generic form (explicit term on HIP + element-wise torch ops), one and many steps
    one_fused = op.solver(w0, 1e-3, op)
    one_generic = op.solver.stepper(w0, 1e-3, op)
    assert rel_l2(one_fused, one_generic) < 1e-13
    wg = w0
    for _ in range(6):
        wg = op.solver.stepper(wg, 1e-3, op)
    monkeypatch.setenv("TCFD_GRAPH", "0")      # every TCFD_* switch is read at plan creation and frozen in the plan
    op.invalidate_plan()
    w6_plain, _ = op(w0, 1e-3, steps=6)
    monkeypatch.setenv("TCFD_GRAPH", "1")
    op.invalidate_plan()
    w6_graph, _ = op(w0, 1e-3, steps=6)
    assert torch.equal(w6_plain, w6_graph) and rel_l2(w6_plain, wg) < 1e-12


@pytest.mark.parametrize("n,tag", [(16, "f64"), (64, "f32"), (256, "f64"), (512, "f32")])
def test_split_and_plain_plans_agree(n, tag, dev, monkeypatch):
    """The radix-2 split of the column transform (default at 1024^2 fp64) can be forced on or off for any
    n >= 16; both plans must give the same step and the same explicit terms."""
    from oracle import ns2d as O

    real = REAL[tag]
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, real)) for s in range(2)]).to(dev)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("TCFD_SPLIT", flag)   # read at plan creation (a new operator = a new plan)
        _, op = build_op(n, tag, "kolmogorov", dev)
        out, dwdt = op(w0, 1e-3, steps=3)
        assert op._plan(w0).info()["split"] == int(flag)
        res[flag] = (out, dwdt, op.explicit_terms(w0))
    monkeypatch.delenv("TCFD_SPLIT")
    # dw/dt = (w_new - w_old) / (3 dt) amplifies the round-off of w by |w| / |w_new - w_old| ~ 1e2
    tols = (1e-13, 1e-11, 1e-12) if tag == "f64" else (5e-7, 2e-5, 2e-6)
    for a, b, tol in zip(res["0"], res["1"], tols):
        assert rel_l2(a, b) < tol


@pytest.mark.parametrize("n", [512, 1024])
def test_cross_lane_column_transforms_agree_with_stockham(n, dev, monkeypatch):
    """512-point column tiles (512^2 and the split 1024^2 plans, fp64) run their transforms with one LDS exchange +
    register<->lane transpositions (xl_col_fft512); TCFD_COLS_XL=0 selects the three-pass Stockham form.  Same
    arithmetic in a different order: steps, explicit terms, residual / stream function agree to round-off."""
    from oracle import ns2d as O

    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 7 + s, torch.float64)) for s in range(2)]).to(dev)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("TCFD_COLS_XL", flag)
        _, op = build_op(n, "f64", "kolmogorov", dev)
        out, dwdt = op(w0, 1e-3, steps=2)
        psi, r = op.stream_and_residual(out, dwdt)
        res[flag] = (out, op.explicit_terms(w0), psi, dwdt, r)
    monkeypatch.delenv("TCFD_COLS_XL")
    t = oracle_tables(n, "f64", "kolmogorov")
    ref, _ = O.advance(w0.cpu(), 1e-3, t, steps=2)
    assert rel_l2(res["1"][0], ref) < 1e-10
    for a, b, tol in zip(res["0"], res["1"], (1e-13, 1e-12, 1e-13, 1e-10, 1e-7)):
        assert rel_l2(a, b) < tol


@pytest.mark.parametrize("n,tag,B", [(64, "f64", 7), (256, "f32", 10), (1024, "f64", 9)])
def test_batch_chunking_is_bit_identical(n, tag, B, dev, monkeypatch):
    """Batched calls run chunk by chunk (cache-sized chunks by default, TCFD_CHUNK forces a size, 0 disables): the
    fields are independent and every chunk runs the same kernels, so results must not change by a single bit --
    steps (one call, several steps, dw/dt), explicit terms and the residual sweep; ragged last chunk included."""
    from oracle import ns2d as O

    real = REAL[tag]
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 50 + s, real)) for s in range(B)]).to(dev)
    res = {}
    for chunk in ("0", "4", "-1"):
        monkeypatch.setenv("TCFD_CHUNK", chunk)
        _, op = build_op(n, tag, "kolmogorov", dev)
        out, dwdt = op(w0, 1e-3, steps=3)
        psi, r = op.stream_and_residual(out, dwdt)
        res[chunk] = (out, dwdt, op.explicit_terms(w0), psi, r)
    monkeypatch.delenv("TCFD_CHUNK")
    for chunk in ("4", "-1"):
        for a, b in zip(res["0"], res[chunk]):
            assert torch.equal(a, b), chunk


@pytest.mark.parametrize("n", [512, 1024])
def test_fp32_column_tile_variants_agree(n, dev, monkeypatch):
    """Round 6: the plane-emitting column passes of the fp32 solver run 8-column cross-lane tiles (TCFD_F32_COLS8=1, default) or
    the 16-column Stockham tiles (=0): the same arithmetic in another order -- a step, dw/dt and the explicit terms agree to fp32
    round-off, and both agree with the fp64 kernels of the same build.  TCFD_NT_OUT=1 (non-temporal stores of the last stage's
    outputs) changes no bit."""
    from oracle import ns2d as O

    B = 3
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, torch.float64)) for s in range(B)])
    res = {}
    for key, env in (("cols8", {"TCFD_F32_COLS8": "1"}), ("cols16", {"TCFD_F32_COLS8": "0"}), ("nt", {"TCFD_NT_OUT": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        _, op = build_op(n, "f32", "kolmogorov", dev)
        out, dwdt = op(w0.to(torch.complex64).to(dev), 1e-3, steps=2)
        res[key] = (out, dwdt, op.explicit_terms(w0.to(torch.complex64).to(dev)))
        for k in env:
            monkeypatch.delenv(k)
    _, op64 = build_op(n, "f64", "kolmogorov", dev)
    ref = op64(w0.to(dev), 1e-3, steps=2)
    for a, b in zip(res["cols8"], res["nt"]):
        assert torch.equal(a, b)
    assert rel_l2(res["cols8"][0], res["cols16"][0]) < 5e-7 and rel_l2(res["cols8"][2], res["cols16"][2]) < 2e-6
    assert scaled_err(res["cols8"][1], res["cols16"][1], res["cols8"][0] / 2e-3) < 1e-6
    for key in ("cols8", "cols16"):
        assert rel_l2(res[key][0], ref[0]) < 2e-6, key


@pytest.mark.parametrize("n,tag", [(64, "f64"), (512, "f64"), (1024, "f64"), (1024, "f32"), (256, "f32")])
def test_row_kernel_variants_agree(n, tag, dev, monkeypatch):
    """Row pass: the register-staged Stockham kernel (TCFD_ROWS_V=5) and the cross-lane kernel (7: 1024 points, fp64 and --
    since round 6 -- fp32; the default there) are the same arithmetic in a different order: explicit terms and a step agree to round-off.  The plan
    reports which kernel it launches; the values of removed kernels (4, 6) mean 5."""
    from oracle import ns2d as O

    real = REAL[tag]
    B = 2
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, real)) for s in range(B)]).to(dev)
    res, kern = {}, {}
    for v in ("5", "7", "4", "0"):
        monkeypatch.setenv("TCFD_ROWS_V", v)
        _, op = build_op(n, tag, "kolmogorov", dev)
        out, _ = op(w0, 1e-3, steps=2)
        kern[v] = op._plan(w0).info()["rows_kernel"]
        res[v] = (out, op.explicit_terms(w0))
    monkeypatch.delenv("TCFD_ROWS_V")
    x7 = 7 if n == 1024 else 5    # cross-lane transforms: 1024 points
    assert kern["5"] == 5 and kern["4"] == 5 and kern["7"] == x7 and kern["0"] == x7
    tols = (1e-13, 1e-12) if tag == "f64" else (5e-7, 2e-6)
    for v in ("7", "4", "0"):
        for a, b, tol in zip(res[v], res["5"], tols):
            assert rel_l2(a, b) < tol, v


def test_dataset_generation_loop_matches_reference_driver_shape(dev, tmp_path):
    """IC -> warm-up -> trajectory -> irfft2 -> bilinear subsample -> dict, checked against the
    oracle running the same loop on the CPU (64^2, 4 samples in 2 batches, 2x subsample)."""
    from oracle import ns2d as O
    from torch_cfd_amd.data_gen import generate_mcwilliams_dataset

    torch.set_default_dtype(torch.float64)
    n, N, bs, dt = 64, 4, 2, 1e-3
    path = str(tmp_path / "mcw.pt")
    data = generate_mcwilliams_dataset(n, N, bs, dt, warmup_steps=5, total_steps=7, record_every_steps=3,
                                       random_state=3, subsample=2, dtype=torch.float32, cdtype=torch.complex128,
                                       device=dev, path=path)
    assert sorted(data) == ["random_states", "residual", "stream", "vort_t", "vorticity"]
    assert data["random_states"].dtype == torch.int32 and data["random_states"].tolist() == [3, 4, 5, 6]
    t = O.make_tables(n, L, 1e-3, 0.0, True, None, torch.float64)
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 3 + s, torch.float64)) for s in range(N)])
    w0, _ = O.advance(w0, dt, t, steps=5)
    ref = O.trajectory(w0, dt, t, num_steps=7, record_every_steps=3, dtype=torch.complex128)
    for k in ("vorticity", "stream", "vort_t", "residual"):
        phys = torch.fft.irfft2(ref[k]).float()
        phys = torch.nn.functional.interpolate(phys, size=(32, 32), mode="bilinear")
        assert data[k].shape == (N, 3, 32, 32) and data[k].dtype == torch.float32 and data[k].device.type == "cpu"
        assert rel_l2(data[k], phys) < (1e-6 if k != "residual" else 1e-4), k
    assert sorted(torch.load(path)) == sorted(data)


@pytest.mark.parametrize("dealias", [0, 1])
def test_legacy_crank_nicolson_step_golden(dealias, dev):
    """imex_crank_nicolson_step / update_residual (fno/data_gen/solvers.py:49-188)."""
    from torch_cfd_amd import solvers

    torch.set_default_dtype(torch.float64)  # like the reference, the mesh is built in the default dtype
    g = load_golden("ns2d_legacy_cn.npz")
    w0 = torch.from_numpy(g["w0"]).to(dev)
    f = torch.from_numpy(g["f"]).to(dev)
    w_next, dwdt, w, psi, res, mesh, lap, filt = solvers.imex_crank_nicolson_step(
        w0, f, 1e-3, 1e-3, diam=L, dealias=bool(dealias), output_rfft=True)
    tag = f"d{dealias}"
    assert rel_l2(w_next, g[f"{tag}_w_next"]) < 1e-12
    assert rel_l2(dwdt, g[f"{tag}_dwdt"]) < 1e-9
    assert rel_l2(psi, g[f"{tag}_psi"]) < 1e-13
    assert rel_l2(res, g[f"{tag}_res"]) < 1e-6
    res2 = solvers.update_residual(w_next, dwdt, f, 1e-3, mesh, lap, dealias_filter=filt, dealias=bool(dealias))
    assert rel_l2(res2, g[f"{tag}_res_update"]) < 1e-9


@pytest.mark.parametrize("tag", ["s1", "s2", "s1_nodealias", "s2_one"])
def test_legacy_crank_nicolson_trajectory_golden(tag, dev):
    """the legacy driver get_trajectory_imex_crank_nicolson (fno/data_gen/solvers.py:268-448)
    against what the imported reference returned: 20 steps, 4 records (every 5th step), subsample 1 and 2, de-aliasing on / off,
    a batch of one (the reference's squeeze())."""
    from torch_cfd_amd import solvers

    torch.set_default_dtype(torch.float64)
    g = load_golden("ns2d_legacy_cn_trajectory.npz")
    w0 = torch.from_numpy(g["w0"]).to(dev)
    f = torch.from_numpy(g["f"]).to(dev)
    if tag == "s2_one":
        w0 = w0[:1]
    out = solvers.get_trajectory_imex_crank_nicolson(w0, f, visc=1e-3, T=0.02, delta_t=1e-3, record_steps=4, diam=L, pbar=False,
                                                     subsample=2 if tag.startswith("s2") else 1, dealias="nodealias" not in tag)
    assert sorted(out) == ["residual", "stream", "t_steps", "vorticity", "vorticity_t"]
    ns = 16 if tag.startswith("s2") else 32
    for k in ("vorticity", "vorticity_t", "stream", "residual"):
        ref = torch.from_numpy(g[f"{tag}_{k}"])
        assert out[k].shape == ref.shape == (w0.shape[0], 4, ns, ns) and out[k].device.type == "cpu" and out[k].dtype == ref.dtype
    assert rel_l2(out["vorticity"], g[f"{tag}_vorticity"]) < 1e-11
    assert rel_l2(out["stream"], g[f"{tag}_stream"]) < 1e-11
    assert rel_l2(out["vorticity_t"], g[f"{tag}_vorticity_t"]) < 1e-8
    # the residual is what is left of O(1) terms that cancel to the de-aliasing error: measured against dw/dt
    assert scaled_err(out["residual"], torch.from_numpy(g[f"{tag}_residual"]), torch.from_numpy(g[f"{tag}_vorticity_t"])) < 1e-9
    assert torch.allclose(out["t_steps"], torch.from_numpy(g[f"{tag}_t_steps"]).to(out["t_steps"].dtype), rtol=1e-6)
    with pytest.raises(IndexError):       # 11 steps, 4 slots, a record every 2: the reference runs past its buffers
        solvers.get_trajectory_imex_crank_nicolson(w0, f, T=0.011, delta_t=1e-3, record_steps=4, diam=L, pbar=False)


def test_legacy_crank_nicolson_trajectory_with_a_forcing_per_sample(dev):
    """The same driver with a (B, n, n) forcing and T / delta_t not an integer (13 steps, records after steps 6 and 12)."""
    from torch_cfd_amd import solvers

    torch.set_default_dtype(torch.float64)
    g = load_golden("ns2d_legacy_cn_trajectory.npz")
    w0 = torch.from_numpy(g["w0"]).to(dev)
    f = torch.from_numpy(g["f"]).to(dev)
    fb = torch.stack([f, 2.0 * f, -0.5 * f])
    out = solvers.get_trajectory_imex_crank_nicolson(w0, fb, visc=1e-3, T=0.0125, delta_t=1e-3, record_steps=2, diam=L, pbar=False,
                                                     subsample=2, dealias=True)
    for k in ("vorticity", "stream"):
        assert out[k].shape == (3, 2, 16, 16) and rel_l2(out[k], g[f"batched_f_{k}"]) < 1e-11, k
    assert rel_l2(out["vorticity_t"], g["batched_f_vorticity_t"]) < 1e-8
    assert scaled_err(out["residual"], torch.from_numpy(g["batched_f_residual"]), torch.from_numpy(g["batched_f_vorticity_t"])) < 1e-9
    assert torch.allclose(out["t_steps"], torch.from_numpy(g["batched_f_t_steps"]).to(out["t_steps"].dtype), rtol=1e-6)


def test_backdiff_golden(dev):
    from torch_cfd_amd import solvers

    torch.set_default_dtype(torch.float64)
    g = load_golden("ns2d_legacy_cn_trajectory.npz")
    x = torch.from_numpy(g["bdf_x"]).to(dev)
    for order in (1, 2, 3, 4, 5):
        assert rel_l2(solvers.backdiff(x, order), g[f"bdf_{order}"]) < 1e-14
    with pytest.raises(NotImplementedError):
        solvers.backdiff(x, 6)


def test_kolmogorov_dataset_golden(dev, tmp_path):
    """the loop of fno/data_gen/data_gen_Kolmogorov2d.py:119-192 (forced operator, per-sample
    filtered-velocity initial condition with the driver's seed rule, warm-up, get_trajectory_imex, irfft2 -> float32 ->
    bilinear subsample, random_states) against the same loop run with the imported reference's components."""
    from torch_cfd_amd.data_gen import generate_kolmogorov_dataset

    torch.set_default_dtype(torch.float64)
    g = load_golden("ns2d_kolmogorov_dataset.npz")
    n, total, batch, seed, sub, warm, steps, every = (int(v) for v in g["params"])
    path = str(tmp_path / "kolmogorov.pt")
    data = generate_kolmogorov_dataset(n, total, batch, 1e-3, warm, steps, every, viscosity=1e-3, peak_wavenumber=4, max_velocity=5,
                                       scale=1.0, random_state=seed, subsample=sub, device=dev, path=path)
    assert sorted(data) == ["random_states", "residual", "stream", "vort_t", "vorticity"]
    assert torch.equal(data["random_states"], torch.from_numpy(g["random_states"])) and data["random_states"].dtype == torch.int32
    for k in ("vorticity", "stream", "vort_t", "residual"):
        assert data[k].shape == g[k].shape == (total, 2, n // sub, n // sub) and data[k].dtype == torch.float32
    # the driver draws sample k of batch i from seed + i + k: sample 1 (batch 0, k = 1) and sample 2 (batch 1, k = 0) coincide
    assert torch.equal(data["vorticity"][1], data["vorticity"][2]) and np.array_equal(g["vorticity"][1], g["vorticity"][2])
    assert rel_l2(data["vorticity"], g["vorticity"]) < 1e-6 and rel_l2(data["stream"], g["stream"]) < 1e-6
    assert scaled_err(data["vort_t"], torch.from_numpy(g["vort_t"]), torch.from_numpy(g["vorticity"]) / 1e-3) < 1e-6
    assert scaled_err(data["residual"], torch.from_numpy(g["residual"]), torch.from_numpy(g["vort_t"])) < 1e-3
    saved = torch.load(path)
    assert all(torch.equal(saved[k], data[k]) for k in data)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_graph_replay_of_interior_steps_is_bit_identical(tag, dev, monkeypatch):
    """steps=k calls on small grids replay a captured hipGraph for the k-2 interior steps (plan-owned stream, event
    fences); the result must be bit-identical to plain launches, across repeated calls and changed dt."""
    n, B = 64, 3
    _, op = build_op(n, tag, "kolmogorov", dev)
    from oracle import ns2d as O
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, REAL[tag])) for s in range(B)]).to(dev)
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("TCFD_GRAPH", flag)
        op.invalidate_plan()               # switches are frozen per plan
        a, da = op(w0, 1e-3, steps=7)
        b, _ = op(a, 5e-4, steps=4)        # different coefficients: the graph is re-captured
        c, _ = op(b, 5e-4, steps=4)        # same key: replayed
        torch.cuda.synchronize()
        outs[flag] = (a.clone(), da.clone(), b.clone(), c.clone())
    for x, y in zip(outs["0"], outs["1"]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("tag,B,steps", [("f64", 5, 3), ("f32", 2, 1), ("f64", 4, 2)])
def test_half_batch_overlap_is_bit_identical(tag, B, steps, dev, monkeypatch):
    """Large problems run the step on two half batches on two streams (row pass of one half beside the column pass of
    the other, TCFD_OVERLAP); batch elements are independent, so the result is bit-identical to the single-stream
    path, odd batch sizes and the dw/dt output included."""
    n = 128
    _, op = build_op(n, tag, "kolmogorov", dev)
    from oracle import ns2d as O
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, REAL[tag])) for s in range(B)]).to(dev)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("TCFD_OVERLAP", flag)
        monkeypatch.setenv("TCFD_GRAPH", "0")
        op.invalidate_plan()
        w, d = op(w0, 1e-3, steps=steps)
        w2, d2 = op(w, 1e-3)
        torch.cuda.synchronize()
        res[flag] = (w.clone(), d.clone(), w2.clone(), d2.clone())
    for x, y in zip(res["0"], res["1"]):
        assert torch.equal(x, y)


def test_c_abi_without_python(dev, tmp_path):
    """examples/c_abi_taylor_green.c: plain C99 + HIP runtime against include/tcfd.h, no Python or PyTorch in the
    process.  Known answer: the Taylor-Green vortex decays like exp(-2 nu k^2 t); the program checks 1e-8 itself."""
    import shutil
    import subprocess
    from conftest import ROOT

    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / ROCm headers")
    import torch_cfd_amd as tc
    tc._lib.load()                                   # makes sure the library is built
    csrc = os.path.join(ROOT, "torch-cfd_amd", "csrc")
    exe = str(tmp_path / "c_abi_taylor_green")
    cmd = [gcc, "-std=c99", "-O2", os.path.join(ROOT, "examples", "c_abi_taylor_green.c"), "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-L" + csrc, "-L/opt/rocm/lib", "-ltcfd_hip", "-lamdhip64", "-lm",
           "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = r.stdout.decode()
    assert r.returncode == 0 and "PASS" in out, out


def test_empty_batch_returns_empty_tensors(dev):
    """Edge case: a (0, n, m) batch goes through like the reference's tensor ops do -- empty results, no launch."""
    _, op = build_op(32, "f64", "kolmogorov", dev)
    w = torch.empty(0, 32, 17, dtype=torch.complex128, device=dev)
    out, dwdt = op(w, 1e-3, steps=3)
    assert out.shape == (0, 32, 17) and dwdt.shape == (0, 32, 17)
    assert op.explicit_terms(w).shape == (0, 32, 17) and op.residual(w, w).shape == (0, 32, 17)
    from torch_cfd_amd import fno
    torch.set_default_dtype(torch.float32)
    with torch.no_grad():
        y = fno.SpectralConvS(2, 3, 4, 4, 3).to(dev)(torch.empty(0, 2, 16, 16, 10, device=dev))
    assert y.shape == (0, 3, 16, 16, 10)


# ----------------------------------------------------------------------------- gradients (differentiable step)
def test_differentiable_step_matches_fused_step_and_cpu_autograd(dev):
    """A state that requires grad steps through torch-cfd_amd/autograd.py (HIP transforms with hand-written adjoints):
    same values as the fused kernels, and the gradient of a scalar loss equals torch's CPU autograd through the oracle's
    torch.fft op sequence (reference: equations.py:139, 285 -- the reference path is differentiable)."""
    from oracle import ns2d as O

    n, B, dt = 32, 2, 1e-3
    _, op = build_op(n, "f64", "kolmogorov", dev)
    t = oracle_tables(n, "f64", "kolmogorov")
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, torch.float64)) for s in range(B)])
    tgt = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 9 + s, torch.float64)) for s in range(B)])

    def loss_of(out, dwdt):
        return (out - tgt.to(out.device)).abs().pow(2).sum() + 1e-6 * dwdt.abs().pow(2).sum()

    fused, fused_dt = op(w0.to(dev), dt, steps=2)                       # no grad: the fused kernels
    wg = w0.to(dev).requires_grad_(True)
    out, dwdt = op(wg, dt, steps=2)                                     # grad: the differentiable path
    assert out.requires_grad and rel_l2(out.detach(), fused) < 1e-12 and rel_l2(dwdt.detach(), fused_dt) < 1e-9
    loss_of(out, dwdt).backward()
    wc = w0.clone().requires_grad_(True)
    ref, ref_dt = O.advance(wc, dt, t, steps=2)
    loss_of(ref, ref_dt).backward()
    assert rel_l2(wg.grad, wc.grad) < 1e-9
    # explicit terms and residual alone
    wg2 = w0.to(dev).requires_grad_(True)
    op.explicit_terms(wg2).abs().pow(2).sum().backward()
    wc2 = w0.clone().requires_grad_(True)
    O.explicit_terms(wc2, t).abs().pow(2).sum().backward()
    assert rel_l2(wg2.grad, wc2.grad) < 1e-10


@pytest.mark.parametrize("n,tag,B,forcing", [(64, "f64", 3, "kolmogorov"), (256, "f32", 2, None), (1024, "f64", 6, "kolmogorov"),
                                            (96, "f64", 2, "sincos"), (160, "f64", 2, None), (512, "f64", 20, None)])
def test_fused_vjp_of_the_explicit_terms_matches_the_tensor_op_path(n, tag, B, forcing, dev, monkeypatch):
    """tcfd_ns2d_explicit_terms_vjp (one row pass with five c2r / four products / four r2c per row pair) against the
    tensor-op backward through the hand-written transform adjoints (TCFD_FUSED_VJP=0): same cotangent of w for a random
    complex cotangent of F; sizes with split plans (1024: the VJP uses whole-column tiles), several chunks per call
    (1024 x 6, 512 x 20), radix-3 / radix-5 grids, fp32."""
    from oracle import ns2d as O

    real = REAL[tag]
    grid, op = build_op(n, tag, forcing, dev)
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 40 + s, real)) for s in range(min(B, 3))])
    w0 = w0.repeat((B + w0.shape[0] - 1) // w0.shape[0], 1, 1)[:B].contiguous()
    w0 = w0 * (1 + 0.1 * torch.arange(B, dtype=real)[:, None, None])
    g = torch.Generator().manual_seed(n)
    cot = torch.complex(torch.randn(B, n, n // 2 + 1, generator=g, dtype=real), torch.randn(B, n, n // 2 + 1, generator=g, dtype=real)).to(dev)
    grads, vals = {}, {}
    for flag in ("1", "0"):
        monkeypatch.setenv("TCFD_FUSED_VJP", flag)
        w = w0.to(dev).requires_grad_(True)
        F = op.explicit_terms(w)
        vals[flag] = F.detach()
        (F * cot.conj()).real.sum().backward()
        grads[flag] = w.grad
    tol = 1e-11 if tag == "f64" else 2e-5
    assert rel_l2(vals["1"], vals["0"]) < (1e-12 if tag == "f64" else 2e-5)
    assert rel_l2(grads["1"], grads["0"]) < tol


@pytest.mark.parametrize("forcing,steps", [("kolmogorov", 1), (None, 2)])
def test_gradient_penalty_through_the_fused_nodes(forcing, steps, dev, monkeypatch):
    """The other second-order use: d/dw of |dL/dw|^2 -- a gradient penalty -- through FusedExplicitTerms and
    StageUpdate (first stage of a step: no previous accumulator; later stages and the second step: with one), fused nodes
    against the tensor-op path."""
    from oracle import ns2d as O

    n, B = 32, 2
    grid, op = build_op(n, "f64", forcing, dev)
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 11 + s, torch.float64)) for s in range(B)]).to(dev)
    pen = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("TCFD_FUSED_VJP", flag)
        monkeypatch.setenv("TCFD_FUSED_STAGE", flag)
        w = w0.clone().requires_grad_(True)
        out, dwdt = op(w, 2e-3, steps=steps)
        loss = out.abs().pow(4).sum() + dwdt.abs().pow(2).sum() * 1e-4
        (gw,) = torch.autograd.grad(loss, w, create_graph=True)
        (gp,) = torch.autograd.grad(gw.abs().pow(2).sum(), w)
        pen[flag] = gp
    assert torch.linalg.norm(pen["0"]) > 0 and rel_l2(pen["1"], pen["0"]) < 1e-9


def test_second_order_gradients_through_the_fused_nodes(dev, monkeypatch):
    """create_graph=True through the fused autograd nodes: a Hessian-vector product of a scalar of one RK4-CN
    step, d/dw <grad_w loss(w), v>, must equal the one of the tensor-op path (TCFD_FUSED_STAGE=0 / TCFD_FUSED_VJP=0), whose
    transforms differentiate through each other; the fused nodes' raw-pointer backward would silently return the gradient of a
    constant."""
    from oracle import ns2d as O

    n, B = 32, 2
    grid, op = build_op(n, "f64", "kolmogorov", dev)
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 7 + s, torch.float64)) for s in range(B)]).to(dev)
    g = torch.Generator().manual_seed(3)
    v = torch.complex(torch.randn(B, n, n // 2 + 1, generator=g, dtype=torch.float64),
                      torch.randn(B, n, n // 2 + 1, generator=g, dtype=torch.float64)).to(dev)
    hvp, first = {}, {}
    for flag in ("1", "0"):
        monkeypatch.setenv("TCFD_FUSED_VJP", flag)
        monkeypatch.setenv("TCFD_FUSED_STAGE", flag)
        w = w0.clone().requires_grad_(True)
        out, _ = op(w, 2e-3, steps=1)
        loss = out.abs().pow(4).sum()                    # a non-quadratic scalar: its Hessian sees the state
        (gw,) = torch.autograd.grad(loss, w, create_graph=True)
        assert gw.requires_grad
        first[flag] = gw.detach()
        (hv,) = torch.autograd.grad((gw * v.conj()).real.sum(), w)
        hvp[flag] = hv
    assert rel_l2(first["1"], first["0"]) < 1e-10
    assert torch.linalg.norm(hvp["0"]) > 0 and rel_l2(hvp["1"], hvp["0"]) < 1e-9


def test_trainable_rk_coefficients_receive_gradients(dev):
    """RK4CrankNicolsonStepper(requires_grad=True): d loss / d gammas from the differentiable path against central
    differences of the FUSED forward step (two different code paths must agree)."""
    import torch_cfd_amd as tc
    from oracle import ns2d as O

    torch.set_default_dtype(torch.float64)
    n, dt = 32, 2e-3
    grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
    op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.1, solver=tc.RK4CrankNicolsonStepper(requires_grad=True)).to(dev)
    w0 = torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 1, torch.float64))[None].to(dev)
    out, _ = op(w0, dt, steps=1)
    assert out.requires_grad
    out.abs().pow(2).sum().backward()
    g = op.solver.params["gammas"].grad.clone()
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0
    fd = torch.zeros_like(g)
    eps = 1e-4
    with torch.no_grad():
        for k in range(len(g)):
            vals = []
            for sgn in (+1, -1):
                op.solver.params["gammas"][k] += sgn * eps
                op._coef_cache = None
                vals.append(op(w0, dt, steps=1)[0].abs().pow(2).sum().item())   # no grad: fused kernels
                op.solver.params["gammas"][k] -= sgn * eps
            fd[k] = (vals[0] - vals[1]) / (2 * eps)
    assert rel_l2(g, fd) < 1e-6


@pytest.mark.parametrize("order,alpha", [(2, 2 / 3), (1.5, 0.5)])
def test_trainable_imex_parameters_receive_gradients(order, alpha, dev):
    """IMEXStepper(requires_grad=True): alpha / beta are 0-dim tensors in the reference's step (equations.py:174-228) and
    receive gradients there; here the differentiable path keeps them attached (stage_schedule(as_tensors=True)).  Checked
    against central differences of the FUSED forward step, with a state that requires grad too (the silent case).  A
    viscous, coarse-step setting (nu dt k^2 ~ 0.5 at the highest modes) so that the loss really depends on the weights of
    the implicit part -- at nu = 1e-3 the dependence drowns in the round-off of the difference quotient."""
    import torch_cfd_amd as tc
    from oracle import ns2d as O

    torch.set_default_dtype(torch.float64)
    n, dt = 32, 2e-2
    grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
    op = tc.NavierStokes2DSpectral(5e-2, grid, drag=0.1, solver=tc.IMEXStepper(order=order, alpha=alpha, beta=0.5,
                                                                                requires_grad=True)).to(dev)
    w0 = torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 1, torch.float64))[None].to(dev).requires_grad_(True)
    scale = 1.0 / w0.detach().abs().pow(2).sum().item()          # loss = |w(2 dt)|^2 / |w(0)|^2 = O(1)
    out, _ = op(w0, dt, steps=2)
    (out.abs().pow(2).sum() * scale).backward()
    names = ("alpha", "beta") if order == 2 else ("alpha",)
    for name in names:
        g = op.solver.params[name].grad
        assert g is not None and torch.isfinite(g).all() and g.abs() > 0, name
        eps, vals = 1e-6, []
        with torch.no_grad():
            for sgn in (+1, -1):
                op.solver.params[name] += sgn * eps
                op._coef_cache = None
                vals.append(op(w0.detach(), dt, steps=2)[0].abs().pow(2).sum().item() * scale)   # no grad: fused kernels
                op.solver.params[name] -= sgn * eps
        fd = (vals[0] - vals[1]) / (2 * eps)
        # round-off floor of the quotient: eps_machine * loss / eps = 1e-16 / 1e-6
        assert abs(g.item() - fd) <= 1e-6 * abs(fd) + 2e-9, (name, g.item(), fd)
    assert w0.grad is not None and torch.isfinite(torch.view_as_real(w0.grad)).all()


def test_user_stepper_subclass_is_not_replaced_by_the_fused_schedule(dev):
    """A subclass that overrides forward() is a different scheme: NavierStokes2DSpectral.forward loops over it."""
    import torch_cfd_amd as tc

    class HalfStep(tc.RK4CrankNicolsonStepper):
        def forward(self, u, dt, equation, params=None):
            return super().forward(u, dt / 2, equation, params)

    _, op = build_op(32, "f64", "kolmogorov", dev)
    g = load_golden("ns2d_trajectory.npz")
    w0 = torch.from_numpy(g["f64_w0"]).to(dev)
    ref, _ = op(w0, 5e-4, steps=2)
    op.solver = HalfStep().to(dev)
    got, dwdt = op(w0, 1e-3, steps=2)
    assert rel_l2(got, ref) < 1e-13 and rel_l2(dwdt, (ref - w0) / 2e-3) < 1e-10


def test_velocity_keeps_the_input_shape_on_composite_grids(dev):
    import torch_cfd_amd as tc

    torch.set_default_dtype(torch.float64)
    for n in (96, 64):
        grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
        for shape in ((n, n // 2 + 1), (2, 3, n, n // 2 + 1)):
            w = torch.randn(*shape, dtype=torch.complex128, device=dev)
            (u, v), psi = tc.vorticity_to_velocity(grid, w)
            assert u.shape == v.shape == psi.shape == w.shape, (n, shape, u.shape)


def test_trajectory_with_require_grad(dev):
    """get_trajectory_imex(require_grad=True) (fno/data_gen/solvers.py:199): same records as without."""
    import torch_cfd_amd as tc

    g = load_golden("ns2d_trajectory.npz")
    _, op = build_op(32, "f64", "kolmogorov", dev)
    w0 = torch.from_numpy(g["f64_w0"]).to(dev)
    a = tc.get_trajectory_imex(op, w0, 1e-3, num_steps=4, record_every_steps=3, dtype=torch.complex128)
    b = tc.get_trajectory_imex(op, w0, 1e-3, num_steps=4, record_every_steps=3, dtype=torch.complex128, require_grad=True)
    for k in a:
        assert not b[k].requires_grad and rel_l2(b[k], a[k]) < 1e-11, k



# ----------------------------------------------------------------------------- grids that are not a power of two
@pytest.mark.parametrize("n,tag,forcing,fused", [
    (96, "f64", "kolmogorov", True), (192, "f64", None, True), (384, "f32", "kolmogorov", True), (768, "f64", "sincos", True),
    (96, "f32", None, True), (768, "f32", None, True),
    (80, "f64", "sincos", True), (160, "f32", "kolmogorov", True), (320, "f64", None, True), (640, "f64", "kolmogorov", True), (640, "f32", None, True),
    (1536, "f64", "kolmogorov", True), (1280, "f32", None, True),
    (48, "f32", None, False), (48, "f64", "sincos", False),
    (100, "f64", "kolmogorov", False), (200, "f32", None, False), (250, "f64", "sincos", False), (36, "f64", None, False)])
def test_grids_with_an_odd_factor_against_oracle(n, tag, forcing, fused, dev):
    """n = p * 2^k (the reference accepts any even n, equations.py:413-422).  n = 3 * 2^k (96 .. 768) and n = 5 * 2^k
    (80 .. 640) run the FUSED column / row kernels (radix-12 / radix-20 first pass: 4-point transforms, twiddles, 3- / 5-point
    transforms in registers); other sizes with a small odd factor (48, 112, ...) the power-of-two HIP transforms composed by
    decimation over the odd factor + the stage loop in tensor ops (mixed_radix.py); every OTHER even n (100, 200, 250, 36:
    odd factor 25, 125, 9 times a power of two below 8) dense device transforms (mixed_radix.DenseDft).  Transforms against
    torch.fft semantics (non-Hermitian c2r input
    included), explicit terms / steps / residual / stream function / trajectory against the oracle."""
    import torch_cfd_amd as tc
    from oracle import ns2d as O

    real, cplx = REAL[tag], CPLX[tag]
    grid, op = build_op(n, tag, forcing, dev)
    t = oracle_tables(n, tag, forcing)
    plan = tc.fft_plan(n, cplx, dev)
    g = torch.Generator().manual_seed(n)
    y = torch.randn(3, n, n, generator=g, dtype=real)
    z = torch.complex(torch.randn(2, n, n // 2 + 1, generator=g, dtype=real), torch.randn(2, n, n // 2 + 1, generator=g, dtype=real))
    ttol = 1e-13 if tag == "f64" else 2e-6
    assert rel_l2(plan.rfft2(y.to(dev)), torch.fft.rfft2(y)) < ttol
    assert rel_l2(plan.irfft2(z.to(dev)), torch.fft.irfft2(z, s=(n, n))) < ttol
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 20 + s, real)) for s in range(2)])
    info = op._plan(w0.to(dev)).info()
    assert ("composite" in info) == (not fused)
    if not fused:
        assert info["dense_dft"] == (0 if n == 48 else 1) and info["composite"] == (3 if n == 48 else 0)
    tol = 1e-10 if tag == "f64" else 4e-6
    assert rel_l2(op.explicit_terms(w0.to(dev)), O.explicit_terms(w0, t)) < (1e-10 if tag == "f64" else 2e-5)
    ref, ref_dt = O.advance(w0, 1e-3, t, steps=3)
    out, out_dt = op(w0.to(dev), 1e-3, steps=3)
    assert out.dtype == cplx and rel_l2(out, ref) < tol
    assert scaled_err(out_dt, ref_dt, ref / 3e-3) < tol
    psi, res = op.stream_and_residual(out, out_dt)
    _, psi_ref = O.stream_and_velocity(ref, t.kx, t.ky)
    assert rel_l2(psi, psi_ref) < tol
    assert scaled_err(res, O.residual(ref, ref_dt, t), ref_dt) < (1e-8 if tag == "f64" else 1e-3)
    traj = tc.get_trajectory_imex(op, w0.to(dev), 1e-3, num_steps=4, record_every_steps=3, dtype=cplx)
    ref_traj = O.trajectory(w0, 1e-3, t, num_steps=4, record_every_steps=3, dtype=cplx)
    assert rel_l2(traj["vorticity"], ref_traj["vorticity"]) < (1e-10 if tag == "f64" else 5e-6)


def test_odd_factor_grid_initial_condition_and_gradients(dev):
    """The device IC generator and the differentiable step on a 96^2 grid (composite transforms underneath)."""
    import torch_cfd_amd as tc
    from oracle import ns2d as O
    from torch_cfd_amd.initial_conditions import vorticity_field

    n = 96
    grid, op = build_op(n, "f64", None, dev, drag=0.0)
    w_dev = vorticity_field(grid, 4, random_state=5, device=dev)
    assert rel_l2(w_dev, O.mcwilliams_vorticity(n, L, 4, 5, torch.float64)) < 1e-11
    t = oracle_tables(n, "f64", None, drag=0.0)
    w0 = torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 6, torch.float64))[None]
    wg = w0.to(dev).requires_grad_(True)
    op(wg, 1e-3)[0].abs().pow(2).sum().backward()
    wc = w0.clone().requires_grad_(True)
    O.advance(wc, 1e-3, t)[0].abs().pow(2).sum().backward()
    assert rel_l2(wg.grad, wc.grad) < 1e-9


@pytest.mark.parametrize("n,tag,split", [(64, "f32", "0"), (128, "f64", "1"), (256, "f32", "0"), (512, "f64", "0"), (1024, "f64", "1")])
def test_packed_nyquist_column_agrees_with_the_lone_tile(n, tag, split, dev, monkeypatch):
    """Pruned plans (2/3-rule mask) carry column n/2 of the planes in the imaginary part of column 0 instead of
    spending a whole column tile on it (TCFD_NYQ_PACK: 0 off, 1 / 2 every pass -- the default --, 3 the opening pass of a
    call only).  The STATE's Nyquist column goes through the same arithmetic either way (exactly equal); everything
    else agrees to round-off (the transforms see Hermitian-averaged inputs).  Non-Hermitian input columns included:
    the test field gets random imaginary parts in its DC / Nyquist columns, which c2r semantics must drop."""
    from oracle import ns2d as O

    real = REAL[tag]
    torch.manual_seed(n)
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, 20 + s, real)) for s in range(3)])
    w0[:, :, 0] += 1j * 0.1 * torch.randn(3, n, dtype=real) * w0[:, :, 0].abs().mean()
    w0[:, :, -1] += (0.05 + 0.1j) * torch.randn(3, n, dtype=real) * w0[:, :, 1].abs().mean()
    w0 = w0.to(dev)
    monkeypatch.setenv("TCFD_SPLIT", split)
    res = {}
    for flag in ("0", "2", "3"):
        monkeypatch.setenv("TCFD_NYQ_PACK", flag)
        _, op = build_op(n, tag, "kolmogorov", dev)
        assert op._plan(w0).info()["keep_cols"] > 0          # pruned: the mode applies
        res[flag] = op(w0, 1e-3, steps=3) + (op(w0, 1e-3)[0],)
    tols = (1e-13, 1e-11, 1e-13) if tag == "f64" else (5e-7, 2e-5, 5e-7)
    for flag in ("2", "3"):
        for a, b, tol in zip(res["0"], res[flag], tols):
            assert rel_l2(a, b) < tol
    # one step: the Nyquist column of the new state only saw element-wise arithmetic
    assert torch.equal(res["0"][2][:, :, -1], res["2"][2][:, :, -1])
    # and against the CPU oracle (which follows torch's c2r semantics for the non-Hermitian columns)
    ref, _ = O.advance(w0.cpu(), 1e-3, oracle_tables(n, tag, "kolmogorov"), steps=3)
    assert rel_l2(res["2"][0], ref) < (1e-11 if tag == "f64" else 2e-5)
