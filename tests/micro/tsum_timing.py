"""The lifting tail's backward (skip = last time slice): t-sum inside the tiled kernel (skip_mode 3) against dL/dz2 written and summed by
tcfd_sum_t_into_last, at the config-5 grid:  python tests/micro/tsum_timing.py 10 16 20 32"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from torch_cfd_amd import fno

dev = torch.device("cuda:0")
res = {}
for width in [int(x) for x in sys.argv[1:]] or [10]:
    for act in ("ReLU", "GELU"):
        torch.manual_seed(0)
        b, X, Y, T = 32, 256, 256, 10
        lin1, lin2 = nn.Conv3d(width, 4 * width, 1).to(dev), nn.Conv3d(4 * width, width, 1).to(dev)
        a = getattr(nn, act)()
        x = torch.randn(b, width, X, Y, T, device=dev)
        s = torch.randn(b, width, X, Y, T, device=dev)
        spec = (True, a, a, 2, None)
        kind = fno._saved_kind(spec, width, 4 * width, width, X * Y * T)
        with torch.no_grad():
            pre = torch.empty_like(x) if kind == 2 else None
            out = fno.hip_pointwise(x, lin1, a, lin2, skip=s, act2=a, skip_last_slice=True, pre=pre)
        kept = out if kind == 1 else pre
        dout = torch.randn_like(out)
        for flag in ("1", "0"):
            os.environ["TCFD_PWB_TSUM"] = flag
            fn = lambda: fno._hip_pointwise_backward(spec, dout, x, s, lin1.weight, lin1.bias, lin2.weight, lin2.bias, None, None, None, None,
                                                     out=kept, compact_skip=True)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                for _ in range(5): fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            res[f"w{width}_{act}_tsum{flag}"] = round(best, 3)
        del x, s, out, dout, pre
        torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
