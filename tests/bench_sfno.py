"""SFNO forward (+ Sobolev loss) at BASELINE config 5: SFNO(24,24,5,width=10,4 layers), x (32,256,256,10) fp32."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_cfd_amd import fno

dev = torch.device("cuda:0")
b, width = int(os.environ.get("B", 32)), int(os.environ.get("WIDTH", 10))
torch.manual_seed(0)
model = fno.SFNO(24, 24, 5, width=width, num_spectral_layers=4).to(dev).eval()
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(b, 256, 256, 10, generator=g).to(dev)
y = torch.randn(b, 256, 256, 10, generator=g).to(dev)
loss_fn = fno.SobolevLoss(n_grid=256, norm_order=0, relative=True).to(dev)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

only_train = os.environ.get("ONLY_TRAIN", "0") == "1"      # a profile of training steps alone (no forward-only loops mixed in)
with torch.no_grad():
    out = model(x)
    loss = loss_fn(out, y)
    t_fwd = t_all = float("nan")
    if not only_train:
        t_fwd = timeit(lambda: model(x))
        t_all = timeit(lambda: loss_fn(model(x), y))
# training step: forward + SobolevLoss + backward (HIP spectral convolutions fwd/bwd, torch pointwise modules under autograd)
model.train()
def train_step():
    model.zero_grad(set_to_none=True)
    loss_fn(model(x), y).backward()
t_train = None
if os.environ.get("TRAIN", "1") == "1":
    torch.cuda.reset_peak_memory_stats()
    t_train = timeit(train_step, 3)
    train_mem = torch.cuda.max_memory_allocated() / 1e9
print(json.dumps({"config": f"SFNO(24,24,5,width={width},layers=4) b={b} 256x256x10 fp32", "forward_ms": round(t_fwd, 2),
                  "forward_plus_loss_ms": round(t_all, 2), "loss": float(loss), "params": sum(p.numel() for p in model.parameters()),
                  "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
                  "train_step_ms": round(t_train, 2) if t_train else None,
                  "train_peak_mem_GB": round(train_mem, 2) if t_train else None}))
