"""Where the host time of the notebook-size training iteration goes (SFNO(32,32,5,width 10), batch 4, 64 x 64 x 10, Adam): cProfile of 30
iterations -- the loop is host-bound (3.6 ms per iteration against 1.8 ms replayed from a graph).  python tests/micro/notebook_train_hostprof.py"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch_cfd_amd import fno

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = fno.SFNO(32, 32, 5, 10, beta=-1e-2).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
loss_fn = fno.SobolevLoss(n_grid=64, norm_order=0, time_average=True, relative=True).to(dev)
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randn(4, 64, 64, 10, generator=g).to(dev)
y = torch.randn(4, 64, 64, 10, generator=g).to(dev)

def it():
    opt.zero_grad(set_to_none=True)
    loss = loss_fn(model(x), y)
    loss.backward()
    opt.step()
    return loss

for _ in range(10): it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): it()
torch.cuda.synchronize()
print("ms per iteration", (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):      # the backward's Python on this thread, where the profiler sees it
    for _ in range(3): it()
    pr.enable()
    for _ in range(30): it()
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(int(os.environ.get("ROWS", 35)))
print("\n".join(l[:170] for l in s.getvalue().splitlines()))
