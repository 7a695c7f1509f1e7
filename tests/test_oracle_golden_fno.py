"""Pin oracle/fno.py against reference-generated vectors (tests/golden/fno_layers.npz). CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import fno as OF


def weights_from_sd(g, key, complex_params=False):
    if complex_params:  # SpectralConv3d: weights1..4 complex (fno3d.py:36-79)
        return [torch.from_numpy(g[f"{key}_sd_weights{i}"]) for i in (1, 2, 3, 4)], None
    w = [torch.view_as_complex(torch.from_numpy(g[f"{key}_sd_weight.{i}"]).contiguous()) for i in range(4)]
    b = None
    if f"{key}_sd_bias.0" in g.files:
        b = [torch.view_as_complex(torch.from_numpy(g[f"{key}_sd_bias.{i}"]).contiguous()) for i in range(4)]
    return w, b


def test_spectral_conv3d():
    g = load_golden("fno_layers.npz")
    w, _ = weights_from_sd(g, "conv3d", complex_params=True)
    y = OF.spectral_conv(torch.from_numpy(g["conv3d_x"]), w, (4, 3, 3))
    assert rel_l2(y, g["conv3d_y"]) < 1e-6


@pytest.mark.parametrize("bias", [0, 1])
def test_spectral_conv_s(bias):
    g = load_golden("fno_layers.npz")
    key = f"convS_bias{bias}"
    w, b = weights_from_sd(g, key)
    assert (b is not None) == bool(bias)
    y = OF.spectral_conv(torch.from_numpy(g[key + "_x"]), w, (4, 3, 3), b, delta=0.5)
    assert rel_l2(y, g[key + "_y"]) < 1e-6


@pytest.mark.parametrize("pad", [0, 1])
@pytest.mark.parametrize("steps", [10, 20, 40])
def test_spectral_conv_t(pad, steps):
    g = load_golden("fno_layers.npz")
    key = f"convT_pad{pad}_s{steps}"
    w, b = weights_from_sd(g, key)
    y = OF.spectral_conv_t(torch.from_numpy(g[key + "_x"]), w, (4, 3, 3), b, delta=0.1, out_steps=steps,
                           temporal_padding=bool(pad))
    assert tuple(y.shape) == g[key + "_y"].shape == (2, 4, 16, 8, steps)
    assert rel_l2(y, g[key + "_y"]) < 1e-6


def test_config5_shaped_layer():
    g = load_golden("fno_layers.npz")
    w, _ = weights_from_sd(g, "convS_c5")
    y = OF.spectral_conv(torch.from_numpy(g["convS_c5_x"]), w, (24, 24, 5))
    assert rel_l2(y, g["convS_c5_y"]) < 1e-6


@pytest.mark.parametrize("order", [0, -1, 1])
@pytest.mark.parametrize("rel", [0, 1])
def test_sobolev_loss(order, rel):
    g = load_golden("fno_layers.npz")
    val = OF.sobolev_loss(torch.from_numpy(g["sob_x"]), torch.from_numpy(g["sob_y"]), 16, norm_order=order,
                          relative=bool(rel))
    assert float(val) == pytest.approx(float(g[f"sob_o{order}_r{rel}"]), rel=1e-5)


@pytest.mark.parametrize("name,modes,delta,kw", [
    ("convS", (4, 3, 3), 0.3, None),
    ("convT_pad", (4, 4, 3), 0.1, {"out_steps": 9, "temporal_padding": True}),
    ("convT_plain", (3, 4, 4), 0.1, {"out_steps": 12, "temporal_padding": False}),
])
def test_gradients_of_the_oracle_match_the_reference(name, modes, delta, kw):
    """Pins the gradient fixtures (tests/golden/fno_grads.npz): autograd through the oracle's torch.fft restatement
    reproduces the reference's input / weight / bias gradients."""
    g = load_golden("fno_grads.npz")
    x = torch.from_numpy(g[name + "_x"]).requires_grad_(True)
    t = torch.from_numpy(g[name + "_t"])
    wr = [torch.from_numpy(g[f"{name}_sd_weight.{k}"]).requires_grad_(True) for k in range(4)]
    has_bias = f"{name}_sd_bias.0" in g.files
    br = [torch.from_numpy(g[f"{name}_sd_bias.{k}"]).requires_grad_(True) for k in range(4)] if has_bias else None
    w = [torch.view_as_complex(a) for a in wr]
    b = [torch.view_as_complex(a) for a in br] if br else None
    if kw is None:
        y = OF.spectral_conv(x, w, modes, b, delta=delta)
    else:
        y = OF.spectral_conv_t(x, w, modes, b, delta=delta, **kw)
    assert rel_l2(y, g[name + "_y"]) < 1e-6
    ((y * t).sum() + 0.5 * (y ** 2).sum()).backward()
    assert rel_l2(x.grad, g[name + "_gx"]) < 1e-6
    for k in range(4):
        assert rel_l2(wr[k].grad, g[f"{name}_g_weight.{k}"]) < 1e-6
        if br:
            assert rel_l2(br[k].grad, g[f"{name}_g_bias.{k}"]) < 1e-6


@pytest.mark.parametrize("steps", [10, 20])
def test_oracle_sfno_matches_reference_golden(steps):
    """oracle/sfno.py (the functional whole-model restatement used as the checker of the full config-5 model on the
    GPU) against the reference's own SFNO output on the tiny model of make_golden.gen_sfno."""
    from oracle import sfno as OS

    g = load_golden("fno_sfno_tiny.npz")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    y = OS.sfno_forward(sd, torch.from_numpy(g["x"]), (4, 4, 3), width=4, num_hidden=2, out_steps=steps)
    assert rel_l2(y, g[f"y{steps}"]) < 2e-6


@pytest.mark.parametrize("tag,width,act", [("w16_gelu", 16, "GELU"), ("w16_relu", 16, "ReLU"), ("w20_gelu", 20, "GELU")])
def test_oracle_sfno_at_the_reference_widths_16_and_20_matches_reference_golden(tag, width, act):
    """The widths / activation the reference trains with besides its default (fno/sfno_pytest.py:258-270, fno/train.py:303,
    its notebooks): the oracle's forward, loss and input gradient (autograd through the restatement) against
    make_golden.gen_grads_wide."""
    from oracle import fno as OF
    from oracle import sfno as OS

    g = load_golden("fno_grads_wide.npz")
    sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "_sd_")}
    x = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
    y = OS.sfno_forward(sd, x, (4, 4, 3), width=width, num_hidden=1, out_steps=10, activation=act)
    assert rel_l2(y, g[f"{tag}_pred"]) < 2e-6
    loss = OF.sobolev_loss(y, torch.from_numpy(g[f"{tag}_target"]), n_grid=16, norm_order=0, relative=True)
    assert float(loss.detach()) == pytest.approx(float(g[f"{tag}_loss"]), rel=1e-5)
    loss.backward()
    assert rel_l2(x.grad, g[f"{tag}_gx"]) < 1e-5


@pytest.mark.parametrize("n", [16, 24])
@pytest.mark.parametrize("steps", [10, 20])
def test_oracle_sfno_with_spatial_padding_matches_reference_golden(n, steps):
    """``SFNO(spatial_padding=8)`` (fno/sfno.py:313-328): the output convolution runs on the zero-framed (n + 16)^2 grid."""
    from oracle import sfno as OS

    g = load_golden("fno_sfno_padding.npz")
    sd = {k[len(f"pad{n}_sd_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"pad{n}_sd_")}
    y = OS.sfno_forward(sd, torch.from_numpy(g[f"pad{n}_x"]), (4, 4, 3), width=4, num_hidden=2, out_steps=steps, spatial_padding=8)
    assert rel_l2(y, g[f"pad{n}_y{steps}"]) < 2e-6


def test_oracle_96_layer_and_resampled_gradients_match_reference_golden():
    """A 96^2 layer, and output + gradients of a layer resampled in space and time (``out_mesh_size``, fno/base.py:229-237)."""
    g = load_golden("fno_sfno_padding.npz")
    w, b = weights_from_sd(g, "c96")
    y = OF.spectral_conv(torch.from_numpy(g["c96_x"]), w, (12, 12, 5), b, delta=0.5)
    assert rel_l2(y, g["c96_y"]) < 1e-6
    for tag in ("up", "down"):
        x = torch.from_numpy(g[f"rs_{tag}_x"]).requires_grad_(True)
        wr = [torch.from_numpy(g[f"rs_{tag}_sd_weight.{k}"]).requires_grad_(True) for k in range(4)]
        y = OF.spectral_conv(x, [torch.view_as_complex(a) for a in wr], (4, 3, 3), out_size=g[f"rs_{tag}_y"].shape[-3:])
        assert rel_l2(y, g[f"rs_{tag}_y"]) < 1e-6
        (y * torch.from_numpy(g[f"rs_{tag}_cot"])).sum().backward()
        assert rel_l2(x.grad, g[f"rs_{tag}_gx"]) < 1e-6
        for k in range(4):
            assert rel_l2(wr[k].grad, g[f"rs_{tag}_g_weight.{k}"]) < 1e-6
