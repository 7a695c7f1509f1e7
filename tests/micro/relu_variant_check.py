import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from torch_cfd_amd import fno
dev = torch.device("cuda:0"); torch.manual_seed(0)
ci, cm, co = 10, 40, 10
lin1, lin2, skc = nn.Conv3d(ci, cm, 1).to(dev), nn.Conv3d(cm, co, 1).to(dev), nn.Conv3d(ci, co, 1).to(dev)
for shape, mode in (((3, ci, 32, 32, 10), 1), ((2, ci, 16, 16, 10), 2)):
    x = torch.randn(*shape, device=dev); s = torch.randn_like(x); dout = torch.randn_like(x)
    spec = (True, nn.ReLU(), nn.ReLU(), mode, None)
    res = {}
    for f in ("1", "0"):
        os.environ["TCFD_PW_BWD_RELU"] = f
        res[f] = fno._hip_pointwise_backward(spec, dout, x, s, lin1.weight, lin1.bias, lin2.weight, lin2.bias,
                                             skc.weight if mode == 1 else None, skc.bias if mode == 1 else None, None, None)
    for k, (a, b) in enumerate(zip(res["1"], res["0"])):
        if a is not None:
            print(mode, k, float((a - b).norm() / b.norm()))
