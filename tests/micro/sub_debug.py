import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd.equations import fft_plan
dev = torch.device("cuda:0")
n, S = 64, 2
x = torch.randn(2, n, n, generator=torch.Generator().manual_seed(1))
xh = torch.fft.rfft2(x).to(dev)
plan = fft_plan(n, torch.complex64, dev)
full = plan.irfft2(xh)
fused = plan.irfft2_subsample(xh, S)
it = torch.nn.functional.interpolate(full.reshape(-1, 1, n, n), size=(n // S, n // S), mode="bilinear").reshape(2, n // S, n // S)
a, b, c, d = full[:, 0::2, 0::2], full[:, 0::2, 1::2], full[:, 1::2, 0::2], full[:, 1::2, 1::2]
hf = 0.5 * (0.5 * a + 0.5 * b) + 0.5 * (0.5 * c + 0.5 * d)
vf = 0.5 * (0.5 * a + 0.5 * c) + 0.5 * (0.5 * b + 0.5 * d)
def df(p, q): return float((p - q).abs().max())
print("fused vs interp", df(fused, it), " fused vs hf", df(fused, hf), " fused vs vf", df(fused, vf), " interp vs hf", df(it, hf), " interp vs vf", df(it, vf))
f64 = 0.25 * (a.double() + b.double() + c.double() + d.double())
print("err vs f64: fused", df(fused.double(), f64), "interp", df(it.double(), f64), "hf", df(hf.double(), f64))
print("mismatch fraction fused vs hf:", float((fused != hf).float().mean()))
for name, cand in {"((a+b)+c)+d": 0.25 * (((a + b) + c) + d), "(a+c)+(b+d)": 0.25 * ((a + c) + (b + d)), "(a+d)+(b+c)": 0.25 * ((a + d) + (b + c)),
                   "((a+c)+b)+d": 0.25 * (((a + c) + b) + d)}.items():
    print(name, float((fused != cand).float().mean()))
for n2, S2 in ((256, 2), (512, 4)):
    x = torch.randn(2, n2, n2, generator=torch.Generator().manual_seed(1))
    xh = torch.fft.rfft2(x).to(dev)
    plan = fft_plan(n2, torch.complex64, dev)
    full = plan.irfft2(xh); fused = plan.irfft2_subsample(xh, S2)
    it = torch.nn.functional.interpolate(full.reshape(-1, 1, n2, n2), size=(n2 // S2, n2 // S2), mode="bilinear").reshape(2, n2 // S2, n2 // S2)
    print(n2, S2, "mismatch fraction", float((fused != it).float().mean()), "max", df(fused, it))
