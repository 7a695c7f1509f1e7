"""micro-benchmark of the plain transform kernels at B=64, 1024^2 (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_cfd_amd as tc
dev = torch.device("cuda:0")
n, B = 1024, 64
for cdt, real in ((torch.complex128, torch.float64), (torch.complex64, torch.float32)):
    plan = tc.fft_plan(n, cdt, dev)
    x = torch.randn(B, n, n, dtype=real, device=dev)
    for _ in range(5):
        xh = plan.rfft2(x)
        y = plan.irfft2(xh)
    torch.cuda.synchronize()
    print(cdt, (y - x).abs().max().item())
