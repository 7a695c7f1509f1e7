"""SFNO config 5 with activation='GELU' (FNO3d's / train.py --activation choice): forward time, run twice -- with the packed
branch-free GELU's wave-uniform small-argument path and (TCFD_PW_ACT_T=0) through the run-time activation switch."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = fno.SFNO(24, 24, 5, width=10, num_spectral_layers=4, activation="GELU").to(dev).eval()
x = torch.randn(32, 256, 256, 10, generator=torch.Generator().manual_seed(0)).to(dev)
scale = float(os.environ.get("XSCALE", 1))
x = x * scale
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    timeit(lambda: model(x), 3)
    t_fwd = timeit(lambda: model(x))
y = torch.randn(32, 256, 256, 10, generator=torch.Generator().manual_seed(1)).to(dev)
loss_fn = fno.SobolevLoss(n_grid=256, norm_order=0, relative=True).to(dev)
model.train()
def train_step():
    model.zero_grad(set_to_none=True)
    loss_fn(model(x), y).backward()
t_train = timeit(train_step, 3) if os.environ.get("TRAIN", "1") == "1" else None
print(json.dumps({"activation": "GELU", "xscale": scale, "act_template": os.environ.get("TCFD_PW_ACT_T", "1"),
                  "forward_ms": round(t_fwd, 3), "train_step_ms": t_train and round(t_train, 2)}))
