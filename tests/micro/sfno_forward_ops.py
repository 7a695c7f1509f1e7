"""ATen operators left in the SFNO config-5 forward + loss (torch.profiler, one call): python tests/micro/sfno_forward_ops.py [rows]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = fno.SFNO(24, 24, 5, width=10, num_spectral_layers=4).to(dev).eval()
x = torch.randn(32, 256, 256, 10, device=dev)
y = torch.randn(32, 256, 256, 10, device=dev)
loss_fn = fno.SobolevLoss(n_grid=256, norm_order=0, relative=True).to(dev)
with torch.no_grad():
    for _ in range(3): loss_fn(model(x), y)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        loss_fn(model(x), y); torch.cuda.synchronize()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=rows, max_name_column_width=70))
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.self_device_time_total > 15:
        chain, p = [], ev.cpu_parent
        while p is not None:
            chain.append(p.name[:30]); p = p.cpu_parent
        print(f"BIG {ev.name[:30]:30s} {ev.self_device_time_total:7.0f} us {str(ev.input_shapes)[:80]:80s} <- {' <- '.join(chain[:3])}")
