"""Pointwise backward at the config-5 grid (b, W, 256, 256, 10) for every width the tiled all-MFMA kernel serves
(csrc/tcfd_fno_tiles.hip): ms per call, issued / useful matrix rate.  Widths 4 / 8 / 10 also run the LDS-staged recomputing kernel
(TCFD_PW_BWD_TILES=0).  usage: pw_bwd_wide_timing.py [widths ...]   (env B = batch, default 32)"""
import json, os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(os.environ.get("B", 32))
MFMA = {4: 18, 8: 34, 10: 59, 12: 75, 14: 81, 16: 88, 20: 196, 24: 244, 32: 352}
res = {}
for W in [int(a) for a in sys.argv[1:]] or [10, 16, 20, 24, 32]:
    ci, cm, co = W, 4 * W, W
    lin1, lin2, skc = nn.Conv3d(ci, cm, 1).to(dev), nn.Conv3d(cm, co, 1).to(dev), nn.Conv3d(ci, co, 1).to(dev)
    x = torch.randn(B, ci, 256, 256, 10, device=dev)
    s = torch.randn_like(x)
    dout = torch.randn_like(x)
    P = 256 * 256 * 10
    for act in os.environ.get("ACTS", "ReLU,GELU").split(","):
        a = getattr(nn, act)()
        spec = (True, a, a, 1, None)
        for tiles in (("1", "0") if W in (4, 8, 10) else ("1",)):
            os.environ["TCFD_PW_BWD_TILES"] = tiles
            kind = fno._saved_kind(spec, ci, cm, co, P)
            with torch.no_grad():
                z2 = torch.empty_like(x) if kind == 2 else None
                y = fno.hip_pointwise(x, lin1, a, lin2, skip=s, skip_conv=skc, act2=a, pre=z2)
            kept = y if kind == 1 else z2
            run = lambda: fno._hip_pointwise_backward(spec, dout, x, s, lin1.weight, lin1.bias, lin2.weight, lin2.bias, skc.weight,
                                                      skc.bias, None, None, out=kept)
            assert run() is not None
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            n_mfma = MFMA.get(W, 0) if tiles == "1" else 0
            useful = 2 * (22 * W * W) * B * P          # MACs per point: 5 products with the 4W x W matrices + 2 with W x W
            res[f"w{W}_{act}_tiles{tiles}"] = {"ms": round(ms, 3), "saved": kind, "mfma_per_16": n_mfma,
                                               "issued_TFLOPs": round(n_mfma * 2048 * (B * P // 16) / ms / 1e9, 1) if n_mfma else None,
                                               "useful_TFLOPs": round(useful / ms / 1e9, 1),
                                               "GBps_6A": round(6 * x.numel() * 4 / ms / 1e6, 1)}
    del x, s, dout
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
