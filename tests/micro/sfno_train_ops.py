"""Which ATen operators (and which autograd nodes) the SFNO config-5 training step still launches: torch.profiler table by
operator, device time per step.  Usage: python tests/micro/sfno_train_ops.py [rows]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = fno.SFNO(24, 24, 5, width=10, num_spectral_layers=4).to(dev).train()
x = torch.randn(32, 256, 256, 10, device=dev)
y = torch.randn(32, 256, 256, 10, device=dev)
loss_fn = fno.SobolevLoss(n_grid=256, norm_order=0, relative=True).to(dev)

def step():
    model.zero_grad(set_to_none=True)
    loss_fn(model(x), y).backward()

step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(); torch.cuda.synchronize()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=rows, max_name_column_width=60))

THR = float(os.environ.get("THR", 100))
if os.environ.get("BIG", "0") == "1":
    # every operator call with more than 100 us of device time, with its shapes and the chain of enclosing operators
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step(); torch.cuda.synchronize()
    for ev in prof.events():
        if ev.device_time_total > THR and ev.cpu_parent is not None or (ev.device_time_total > THR and ev.name.startswith("aten::")):
            chain, p = [], ev.cpu_parent
            while p is not None:
                chain.append(p.name[:40]); p = p.cpu_parent
            if ev.self_device_time_total > THR:
                print(f"{ev.name[:40]:40s} {ev.self_device_time_total:8.0f} us  {str(ev.input_shapes)[:90]:90s} <- {' <- '.join(chain[:4])}")
