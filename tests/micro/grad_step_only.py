"""Differentiable RK4-CN steps only (1024^2 x 8 fp64, fused nodes), for a kernel profile: bash tests/micro/prof_cmd.sh grad 30 python tests/micro/grad_step_only.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch_cfd_amd as tc
from torch_cfd_amd.initial_conditions import vorticity_field
dev = torch.device("cuda:0")
torch.set_default_dtype(torch.float64)
n, B, L = 1024, 8, 2 * math.pi
grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.1, forcing_fn=tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=4),
                               solver=tc.RK4CrankNicolsonStepper()).to(dev)
w0 = tc.fft_plan(n, torch.complex128, dev).rfft2(vorticity_field(grid, 4, batch_seeds=list(range(B)), device=dev))
for _ in range(int(os.environ.get("REPS", 6))):
    w = w0.detach().requires_grad_(True)
    op(w, 1e-3)[0].abs().pow(2).sum().backward()
torch.cuda.synchronize()
