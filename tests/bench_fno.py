"""Spectral-convolution layer timing at BASELINE config 5 shape (b=32, width 10, 256x256x10, modes 24/24/5).
Prints per-call ms of the HIP layer (torch.cuda.Event on the current stream, which the library launches on),
algorithmic GB/s (2.3 GB per layer) and, for context, the reference-style op stream run with
torch.fft on the same GPU (full rfftn, zero-filled spectrum, 4 einsums, irfftn)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_cfd_amd import fno

dev = torch.device("cuda:0")
b, C, X, Y, T, modes = int(os.environ.get("B", 32)), int(os.environ.get("WIDTH", 10)), 256, 256, 10, (24, 24, 5)
torch.manual_seed(0)
layer = fno.SpectralConvS(C, C, *modes).to(dev)
x = torch.randn(b, C, X, Y, T, device=dev)

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def ref_style(v):
    vh = torch.fft.rfftn(v, dim=(-3, -2, -1))
    out = torch.zeros(b, C, X, Y, T // 2 + 1, dtype=vh.dtype, device=dev)
    mx, my, mt = modes
    sx = [slice(0, mx), slice(-mx, None)]; sy = [slice(0, my), slice(-my, None)]
    for ix in range(2):
        for iy in range(2):
            w = torch.view_as_complex(layer.weight[ix + 2 * iy])
            out[..., sx[ix], sy[iy], :mt] = torch.einsum("bixyt,ioxyt->boxyt", vh[..., sx[ix], sy[iy], :mt], w)
    return torch.fft.irfftn(out, s=(X, Y, T), dim=(-3, -2, -1))

with torch.no_grad():
    y = layer(x)
    yr = ref_style(x)
    err = ((y - yr).norm() / yr.norm()).item()
    t_hip = timeit(lambda: layer(x))
    t_hip_valu = timeit(lambda: fno.hip_spectral_conv(x, list(layer.weight), None, 1.0, modes, use_mfma=False))
    t_ref = timeit(lambda: ref_style(x), 5)
AH = b * C * X * Y * T * 4
Q = 2 * modes[1] * modes[2]
algo = 2 * AH + 4 * b * C * X * Q * 8 + 4 * b * C * 2 * modes[0] * Q * 8
print(json.dumps({"shape": [b, C, X, Y, T], "modes": modes, "hip_ms": round(t_hip, 3), "hip_valu_contract_ms": round(t_hip_valu, 3),
                  "torch_fft_opstream_ms": round(t_ref, 3), "algo_GB": round(algo / 1e9, 3),
                  "algo_GBps": round(algo / 1e9 / (t_hip * 1e-3), 1), "rel_l2_vs_torch_fft": err}))
