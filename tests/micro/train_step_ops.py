"""Which torch ops (not library kernels) run inside one SFNO config-5 training step, with the Python frames that issued them:
python tests/micro/train_step_ops.py   (torch.profiler, one step after warm-up)"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = fno.SFNO(24, 24, 5, width=int(os.environ.get("WIDTH", 10)), num_spectral_layers=4).to(dev).train()
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(32, 256, 256, 10, generator=g).to(dev)
y = torch.randn(32, 256, 256, 10, generator=g).to(dev)
loss_fn = fno.SobolevLoss(n_grid=256, norm_order=0, relative=True).to(dev)

def train_step():
    model.zero_grad(set_to_none=True)
    loss_fn(model(x), y).backward()
for _ in range(2): train_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    train_step()
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue                                     # count leaves only
    frames = [f for f in (ev.stack or []) if "torch_cfd_amd" in f or "torch-cfd_amd" in f]
    where = frames[0].split("/")[-1] if frames else ("autograd engine" if not ev.stack else ev.stack[0].split("/")[-1])
    key = (ev.name, str(ev.input_shapes)[:80], where[:70])
    d = rows.setdefault(key, [0, 0.0])
    d[0] += 1; d[1] += ev.device_time_total
tot = 0
for k, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("ROWS", 45))]:
    print(f"{t:8.1f} us {n:3d}x  {k[0]:28s} {k[1]:80s} {k[2]}")
print("all aten leaves with device time:", round(sum(t for _, t in rows.values()) / 1e3, 3), "ms")
