"""SFNO(24,24,5,width 10).double() forward on (B,256,256,10): ms per forward beside the fp32 model (B from argv, default 8)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
m32 = fno.SFNO(24, 24, 5, width=10, num_spectral_layers=4).to(dev).eval()
m64 = fno.SFNO(24, 24, 5, width=10, num_spectral_layers=4).to(dev).eval()
m64.load_state_dict(m32.state_dict()); m64 = m64.double()
x = torch.randn(B, 256, 256, 10, device=dev)
def timed(m, inp, n=5):
    with torch.no_grad():
        m(inp); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): m(inp)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t32, t64 = timed(m32, x), timed(m64, x.double())
print(json.dumps({"batch": B, "fp32_ms": round(t32, 3), "fp64_ms": round(t64, 3), "ratio": round(t64 / t32, 2)}))
