import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fno as OF
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
for n, b, nt, real in ((80, 3, 4, torch.float64), (96, 2, 4, torch.float64), (160, 2, 4, torch.float64), (64, 2, 4, torch.float64), (640, 1, 2, torch.float64), (80, 3, 4, torch.float32)):
    g = torch.Generator().manual_seed(n + nt)
    x = torch.randn(b, n, n, nt, generator=g, dtype=real)
    y = x + 0.3 * torch.randn(b, n, n, nt, generator=g, dtype=real)
    for kw in (dict(norm_order=0, relative=True), dict(norm_order=-1, relative=False), dict(norm_order=1, relative=True, time_average=False)):
        loss = fno.SobolevLoss(n_grid=n, **kw).to(dev)
        ref = float(OF.sobolev_loss(x.double(), y.double(), n, **kw))
        fused = float(loss(x.to(dev), y.to(dev)))
        os.environ["TCFD_LOSS_FUSED"] = "0"
        comp = float(loss(x.to(dev), y.to(dev)))
        del os.environ["TCFD_LOSS_FUSED"]
        print(n, real, kw, "fused rel err %.2e  composed rel err %.2e" % (abs(fused - ref) / abs(ref), abs(comp - ref) / abs(ref)))
