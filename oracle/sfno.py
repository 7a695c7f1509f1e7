"""CPU oracle: the whole SFNO forward as a FUNCTION of a ``state_dict`` (no modules, no HIP).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Restates, in plain torch-CPU ops, what
``SFNO.forward`` of the reference computes for out_dim = 1 (fno/sfno.py:596-620):

  lifting (sfno.py:196-262)   v + PE (sfno.py:25-113)  ->  LayerNormnd (= GroupNorm, 1 group, eps 1e-7; base.py:61-82)
                              ->  1x1x1 projection  ->  w = FFN(SpectralConvT(v))  ->  act(v[..., -1:] + w)
  hidden layers (:607-614)    v <- act(FFN(SpectralConvS(v)) + W v)
  reduction (:616)            1x1x1 convolution to one channel
  output (sfno.py:265-328)    prepend the last input frame, SpectralConvT with left zero padding by T and bias,
                              ``v_res[..., -1:] + conv(...)[..., -out_steps:]``

Pinned by ``tests/test_oracle_golden_fno.py::test_oracle_sfno_matches_reference_golden`` against the
reference's own output on the tiny model of ``tests/golden/fno_sfno_tiny.npz`` (made by importing the
reference, ``tests/golden/make_golden.py``).  The GPU tests use it as the checker of the full BASELINE
config-5 model on a batch slice.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F

from .fno import spectral_conv, spectral_conv_t

_ACT = {"ReLU": F.relu, "GELU": F.gelu, "SiLU": F.silu, "Tanh": torch.tanh, "Identity": lambda z: z}


def positional_table(nx: int, ny: int, nt: int, channels: int, beta: float, max_time_steps: int = 100) -> torch.Tensor:
    """(1, channels, nx, ny, nt): x, y, t coordinates, then e^{beta t} sin / cos(pi (k+1) t) (sfno.py:91-106)."""
    gx, gy = torch.linspace(0, 1, nx), torch.linspace(0, 1, ny)
    gt = torch.linspace(0, 1, max_time_steps + 1)[1: nt + 1]
    X, Y, T = torch.meshgrid(gx, gy, gt, indexing="ij")
    rows = [X, Y, T]
    for k in range(channels - 3):
        wave = torch.sin if k % 2 == 0 else torch.cos
        rows.append((torch.exp(beta * gt) * wave(torch.pi * (k + 1) * gt)).reshape(1, 1, nt).repeat(nx, ny, 1))
    return torch.stack(rows).unsqueeze(0)


def _complex_blocks(sd: Dict[str, torch.Tensor], prefix: str, name: str):
    """The four corner blocks ``prefix.name.0..3`` stored as real (..., 2) tensors (base.py:139-155)."""
    return [torch.view_as_complex(sd[f"{prefix}.{name}.{i}"].contiguous()) for i in range(4)]


def _conv1(v: torch.Tensor, sd, prefix: str) -> torch.Tensor:
    """1x1x1 Conv3d ``prefix`` applied point-wise: weight (Co, Ci, 1, 1, 1), bias (Co,)."""
    w = sd[f"{prefix}.weight"].flatten(1)
    return torch.einsum("oc,bcxyt->boxyt", w, v) + sd[f"{prefix}.bias"].view(1, -1, 1, 1, 1)


def _ffn(v, sd, prefix, act):
    """PointwiseFFN: linear1 -> activation -> linear2 (base.py:86-111)."""
    return _conv1(act(_conv1(v, sd, f"{prefix}.linear1")), sd, f"{prefix}.linear2")


def sfno_forward(sd: Dict[str, torch.Tensor], v: torch.Tensor, modes: Sequence[int], width: int, num_hidden: int,
                 out_steps: int, latent_steps: int = 10, beta: float = -1e-2, delta: float = 1e-1,
                 activation: str = "ReLU", norm: str = "backward", lift_activation: bool = True,
                 spatial_padding: int = 0) -> torch.Tensor:
    """(b, x, y, t_in) -> (b, x, y, out_steps); ``sd`` = the model's ``state_dict`` on the CPU,
    ``num_hidden`` = num_spectral_layers - 1."""
    act = _ACT[activation]
    v_res = v
    v = v.unsqueeze(1)
    nx, ny, nt = v.shape[-3:]
    # ---- lifting operator
    v = v + positional_table(nx, ny, nt, width, beta).to(v.dtype)
    v = F.group_norm(v, 1, sd["lifting_operator.norm.weight"], sd["lifting_operator.norm.bias"], eps=1e-7)
    v = _conv1(v, sd, "lifting_operator.proj")
    w = spectral_conv_t(v, _complex_blocks(sd, "lifting_operator.sconv", "weight"), modes, None, delta,
                        out_steps=latent_steps, temporal_padding=False, norm=norm)
    if lift_activation:
        v = act(v[..., -1:] + _ffn(w, sd, "lifting_operator.mlp", act))
    else:
        v = v[..., -1:] + _conv1(w, sd, "lifting_operator.mlp")
    # ---- hidden layers
    for layer in range(num_hidden):
        x1 = spectral_conv(v, _complex_blocks(sd, f"spectral_conv.{layer}", "weight"), modes, None, 1.0, norm)
        v = act(_ffn(x1, sd, f"mlp.{layer}", act) + _conv1(v, sd, f"w.{layer}"))
    v = _conv1(v, sd, "reduction")
    # ---- output operator (out_dim = 1: no Helmholtz projection)
    frames = torch.cat([v_res.unsqueeze(1)[..., -1:], v], dim=-1)
    sp = spatial_padding
    if sp > 0:      # fno/sfno.py:313-321: zero frame of sp points around the spatial grid, cropped again after the convolution
        frames = F.pad(frames, (0, 0, sp, sp, sp, sp))
    out = spectral_conv_t(frames, _complex_blocks(sd, "output_operator.conv", "weight"), modes,
                          _complex_blocks(sd, "output_operator.conv", "bias"), delta, out_steps=out_steps + 1,
                          temporal_padding=True, norm=norm)
    if sp > 0:
        out = out[..., sp:-sp, sp:-sp, :]
    return (v_res.unsqueeze(1)[..., -1:] + out[..., -out_steps:]).squeeze(1)
