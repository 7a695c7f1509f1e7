"""Random shapes through the tiled pointwise kernels (csrc/tcfd_fno_tiles.hip): backward against float64 autograd of the einsum
form, forward (widths 16 ... 32, TCFD_PW_FWD_TILES=2) against float64 modules.  Point counts that are multiples of 4 but not of 16,
tiny ones, batch 1, every mode and activation.  Prints the worst relative errors; exits non-zero on a failure."""
import os, random, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["TCFD_PW_FWD_TILES"] = "2"
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
rng = random.Random(int(os.environ.get("SEED", 0)))
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
worst = {}
n_cases = int(os.environ.get("CASES", 120))
for case in range(n_cases):
    W = rng.choice([4, 6, 8, 10, 12, 14, 16, 20, 24, 32])
    mode = rng.choice([0, 1, 2])
    act = rng.choice(["ReLU", "GELU", "SiLU", "Tanh", None])
    b = rng.choice([1, 2, 3, 5])
    X, Y, T = rng.choice([(1, 2, 2), (2, 2, 1), (3, 4, 2), (5, 4, 6), (7, 4, 10), (4, 4, 10), (6, 10, 2), (9, 12, 3), (16, 16, 10), (2, 2, 3)])
    if (X * Y * T) % 4:
        continue
    torch.manual_seed(case)
    lin1, lin2 = nn.Conv3d(W, 4 * W, 1).to(dev), nn.Conv3d(4 * W, W, 1).to(dev)
    skc = nn.Conv3d(W, W, 1).to(dev) if mode == 1 else None
    a1 = getattr(nn, act)() if act else None
    x = torch.randn(b, W, X, Y, T, device=dev, requires_grad=True)
    s = None
    if mode == 1:
        s = torch.randn(b, W, X, Y, T, device=dev, requires_grad=True)
    elif mode == 2:
        s = torch.randn(b, W, X, Y, rng.choice([1, 3, T]), device=dev, requires_grad=True)
    out = fno.hip_pointwise(x, lin1, a1, lin2, skip=s, skip_conv=skc, act2=a1, skip_last_slice=(mode == 2))
    assert out is not None and out.grad_fn is not None, (W, mode, act)
    fell = []
    saved = fno._pointwise_reference
    fno._pointwise_reference = lambda *a, **k: (fell.append(1), saved(*a, **k))[1]
    t = torch.randn_like(out)
    (out * t).sum().backward()
    fno._pointwise_reference = saved
    d = lambda v: v.detach().double().requires_grad_(True) if v is not None else None
    leaves = [d(x), d(s), d(lin1.weight), d(lin1.bias), d(lin2.weight), d(lin2.bias), d(skc.weight) if skc else None, d(skc.bias) if skc else None, None, None]
    ref_out = fno._pointwise_reference((True, a1, a1, mode, None), *leaves)
    (ref_out * t.double()).sum().backward()
    errs = {"out": rel(out, ref_out), "dx": rel(x.grad, leaves[0].grad), "dw1": rel(lin1.weight.grad, leaves[2].grad), "db1": rel(lin1.bias.grad, leaves[3].grad),
            "dw2": rel(lin2.weight.grad, leaves[4].grad), "db2": rel(lin2.bias.grad, leaves[5].grad)}
    if s is not None:
        errs["ds"] = rel(s.grad, leaves[1].grad)
    if skc is not None:
        errs["dws"] = rel(skc.weight.grad, leaves[6].grad); errs["dbs"] = rel(skc.bias.grad, leaves[7].grad)
    bad = {k: v for k, v in errs.items() if not (v < 5e-5)}
    if bad or fell:
        print("FAIL", dict(W=W, mode=mode, act=act, b=b, shape=(X, Y, T), fell_back=len(fell)), bad)
        sys.exit(1)
    for k, v in errs.items():
        worst[k] = max(worst.get(k, 0.0), v)
print("cases", n_cases, "worst", {k: f"{v:.2e}" for k, v in worst.items()})
