"""Where does one iteration of the reference's notebook training loop go (SFNO(32,32,5,width 10), batch 4, 64x64x10, Adam)?
torch.profiler by operator + wall time split (host-bound or device-bound?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = fno.SFNO(32, 32, 5, 10, beta=-1e-2).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
loss_fn = fno.SobolevLoss(n_grid=64, norm_order=0, time_average=True, relative=True).to(dev)
x = torch.randn(4, 64, 64, 10, device=dev); y = torch.randn(4, 64, 64, 10, device=dev)
def it():
    opt.zero_grad(set_to_none=True)
    loss = loss_fn(model(x), y); loss.backward(); opt.step(); return loss
for _ in range(5): it()
torch.cuda.synchronize()
def timed(fn, n=30):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    host = time.perf_counter() - t
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, host / n * 1e3
print("iteration ms (wall, host-enqueue):", timed(it))
def fwd():
    with torch.no_grad(): model(x)
print("forward no-grad:", timed(fwd))
def fwd_bwd():
    model.zero_grad(set_to_none=True); loss_fn(model(x), y).backward()
print("fwd+loss+bwd:", timed(fwd_bwd))
it(); torch.cuda.synchronize()
def adam(): opt.step()
print("adam step:", timed(adam))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    it(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=int(sys.argv[1]) if len(sys.argv) > 1 else 28, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
# who issues the small copies / fills / host-to-device transfers?
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof2:
    it(); torch.cuda.synchronize()
from collections import Counter
cnt = Counter()
for ev in prof2.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::to", "aten::_to_copy", "aten::tensor", "aten::scalar_tensor", "aten::zeros", "aten::full", "aten::item", "aten::_local_scalar_dense", "aten::empty", "aten::contiguous", "aten::clone"):
        st = [f for f in (ev.stack or []) if "torch-cfd_amd" in f or "torch_cfd_amd" in f or "optim" in f]
        cnt[(ev.name, st[0][-70:] if st else "?")] += 1
for (name, where), c in cnt.most_common(45):
    print(f"{c:4d} {name:28s} {where}")
