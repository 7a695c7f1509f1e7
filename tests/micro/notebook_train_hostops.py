"""Which lines of the package issue the small host-side operations of one training iteration (notebook config)?
A TorchDispatchMode logs every ATen op with the innermost repo frame that called it."""
import os, sys, traceback, torch
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = fno.SFNO(32, 32, 5, 10, beta=-1e-2).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
loss_fn = fno.SobolevLoss(n_grid=64, norm_order=0, time_average=True, relative=True).to(dev)
x = torch.randn(4, 64, 64, 10, device=dev); y = torch.randn(4, 64, 64, 10, device=dev)
def it():
    opt.zero_grad(set_to_none=True)
    loss = loss_fn(model(x), y); loss.backward(); opt.step(); return loss
for _ in range(3): it()
cnt = Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        frames = [f for f in traceback.extract_stack() if "torch_cfd_amd" in f.filename or "torch-cfd_amd" in f.filename or "optim/" in f.filename]
        where = f"{os.path.basename(frames[-1].filename)}:{frames[-1].lineno} {frames[-1].name}" if frames else "?"
        h2d = ""
        if "_to_copy" in name or name.startswith("copy_"):
            try:
                src = args[1] if name.startswith("copy_") else args[0]
                dst_dev = (args[0].device if name.startswith("copy_") else (kwargs or {}).get("device", src.device))
                if src.device.type == "cpu" and torch.device(dst_dev).type == "cuda": h2d = " H2D"
                if src.device.type == "cuda" and torch.device(dst_dev).type == "cpu": h2d = " D2H"
            except Exception: pass
        if "_local_scalar_dense" in name and args[0].device.type == "cuda": h2d = " SYNC"
        cnt[(name + h2d, where)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    it()
torch.cuda.synchronize()
tot = sum(cnt.values())
print("ops per iteration:", tot)
for (name, where), c in cnt.most_common(70):
    print(f"{c:4d} {name:36s} {where}")
