"""Pointwise backward at the config-5 activation size (32, 10, 256, 256, 10): ms per call for each kernel variant
(TCFD_PW_BWD = 5 all-MFMA, 2 / 4 LDS-staged with that many waves per 64 points, 1 one wave per 64 points)."""
import json, os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
torch.manual_seed(0)
ci, cm, co = 10, 40, 10
lin1, lin2, skc = nn.Conv3d(ci, cm, 1).to(dev), nn.Conv3d(cm, co, 1).to(dev), nn.Conv3d(ci, co, 1).to(dev)
x = torch.randn(32, ci, 256, 256, 10, device=dev)
s = torch.randn_like(x)
dout = torch.randn_like(x)
res = {}
for act in os.environ.get("ACTS", "ReLU,GELU").split(","):
    spec = (True, getattr(nn, act)(), getattr(nn, act)(), 1, None)
    for flag in sys.argv[1:] or ["5", "2", "4"]:
        os.environ["TCFD_PW_BWD"] = flag
        # WITH_OUT=1 (default): the block's forward output is handed over, as the autograd nodes do (ReLU: the 71-product kernel)
        y = None
        if os.environ.get("WITH_OUT", "1") == "1":
            with torch.no_grad():
                y = fno.hip_pointwise(x, lin1, spec[1], lin2, skip=s, skip_conv=skc, act2=spec[2])
        run = lambda: fno._hip_pointwise_backward(spec, dout, x, s, lin1.weight, lin1.bias, lin2.weight, lin2.bias, skc.weight, skc.bias,
                                                  None, None, out=y)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        res[f"{act}_mode{flag}_ms"] = round(e0.elapsed_time(e1) / 5, 3)
print(json.dumps(res))
