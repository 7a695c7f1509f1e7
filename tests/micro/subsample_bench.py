"""Output side of one C4 record (4 fields x 64 samples, 512^2 complex64 -> 256^2 float32): fused c2r + subsample against the
two-step path (HIP irfft2 + F.interpolate)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd.data_gen import spectral_to_physical
dev = torch.device("cuda:0")
xh = torch.fft.rfft2(torch.randn(4, 64, 512, 512, device=dev))
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = {}
for fused in ("1", "0"):
    os.environ["TCFD_FUSED_SUBSAMPLE"] = fused
    out["fused_ms" if fused == "1" else "two_step_ms"] = round(timeit(lambda: spectral_to_physical(xh, 256, torch.float32)), 4)
print(json.dumps(out))
