"""Forward pointwise block (FFN + skip convolution + activation) at the config-5 grid: matrix-pipe tile kernel (TCFD_PW_FWD_TILES=2)
against the packed vector kernel (=0):  python tests/micro/pw_fwd_tiles_timing.py 16 20 24 32   (env ACT=ReLU|GELU)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from torch_cfd_amd import fno

dev = torch.device("cuda:0")
res = {}
act = os.environ.get("ACT", "ReLU")
for width in [int(x) for x in sys.argv[1:]] or [16, 20, 24, 32]:
    torch.manual_seed(0)
    b, X, Y, T = 32, 256, 256, 10
    lin1, lin2, skc = nn.Conv3d(width, 4 * width, 1).to(dev), nn.Conv3d(4 * width, width, 1).to(dev), nn.Conv3d(width, width, 1).to(dev)
    a = getattr(nn, act)()
    x = torch.randn(b, width, X, Y, T, device=dev)
    s = torch.randn(b, width, X, Y, T, device=dev)
    out = torch.empty_like(x)
    for flag in ("2", "0"):
        os.environ["TCFD_PW_FWD_TILES"] = flag
        with torch.no_grad():
            fn = lambda: fno.hip_pointwise(x, lin1, a, lin2, skip=s, skip_conv=skc, act2=a, out=out)
            assert fn() is not None
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                for _ in range(5): fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
        res[f"w{width}_{act}_tiles{flag}"] = round(best, 3)
    del x, s, out
    torch.cuda.empty_cache()
print(json.dumps(res))
