"""One differentiable RK4-CN step (forward + backward of a scalar loss w.r.t. the state) on the device: ms with the fused
explicit-terms VJP (default) and with the tensor-op path (TCFD_FUSED_VJP=0), next to the forward-only fused step."""
import json, math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch_cfd_amd as tc
from torch_cfd_amd.initial_conditions import vorticity_field
dev = torch.device("cuda:0")
torch.set_default_dtype(torch.float64)
res = {}
for n, B in ((512, 16), (1024, 8)):
    L = 2 * math.pi
    grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
    op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.1, forcing_fn=tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=4),
                                   solver=tc.RK4CrankNicolsonStepper()).to(dev)
    w0 = tc.fft_plan(n, torch.complex128, dev).rfft2(vorticity_field(grid, 4, batch_seeds=list(range(B)), device=dev))
    def fwd_only():
        with torch.no_grad(): op(w0, 1e-3)
    def grad_step():
        w = w0.detach().requires_grad_(True)
        op(w, 1e-3)[0].abs().pow(2).sum().backward()
    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
    r = {"forward_only_ms": round(timed(fwd_only), 3)}
    for name, vjp, stage in (("fwd_bwd_fused_nodes_ms", "1", "1"), ("fwd_bwd_fused_vjp_tensor_stage_ms", "1", "0"),
                             ("fwd_bwd_tensor_ops_ms", "0", "0")):
        os.environ["TCFD_FUSED_VJP"], os.environ["TCFD_FUSED_STAGE"] = vjp, stage
        r[name] = round(timed(grad_step), 3)
    res[f"{n}x{B}_f64"] = r
print(json.dumps(res))
