"""SpectralConvS layer (b=32, 10 -> 10 channels, T=10, modes 24/24/5... capped by the grid) on grids off the fused kernels
(dense pruned transforms, thin GEMMs) next to power-of-two neighbours (fused kernels): ms per layer and ns per grid point."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = {}
for X in (64, 80, 96, 128, 160, 192, 256, 272, 320, 384):
    m = min(24, X // 4)
    layer = fno.SpectralConvS(10, 10, m, m, 5).to(dev)
    b = 32 if X <= 256 else 8
    x = torch.randn(b, 10, X, X, 10, device=dev)
    with torch.no_grad():
        t = timeit(lambda: layer(x))
    res[X] = {"ms": round(t, 3), "ns_per_point": round(t * 1e6 / (b * 10 * X * X * 10), 4), "batch": b, "path": ("dense GEMMs" if os.environ.get("TCFD_FNO_DENSE") == "1" and X & (X - 1) else
                       "direct-DFT kernels" if os.environ.get("TCFD_FNO_DFT") == "1" or (X & (X - 1) and X % 3 and X % 5) else "fft kernels")}
print(json.dumps(res))
