import sys, time, math, torch
sys.path.insert(0, '/root/repo')
import torch_cfd_amd as tc
from torch_cfd_amd.initial_conditions import vorticity_field
dev = torch.device('cuda'); torch.set_default_dtype(torch.float32)
n, B, L = 256, 16, 2 * math.pi
grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.0, smooth=True, solver=tc.RK4CrankNicolsonStepper()).to(dev)
with torch.no_grad():
    w = tc.fft_plan(n, torch.complex64, dev).rfft2(torch.cat([vorticity_field(grid, 4, batch_seeds=list(range(i, i + 8)), device=dev) for i in (0, 8)]))
    w = op(w, 1e-3, steps=400)[0]; torch.cuda.synchronize()
    t = time.perf_counter(); w = op(w, 1e-3, steps=400)[0]; torch.cuda.synchronize(); el = time.perf_counter() - t
print("C2 steps/s %.1f  ms/step %.4f" % (400 / el, el / 400 * 1e3))
