"""GPU: BASELINE config-4 shard (512^2 x 64 fp64, unforced): per-call and fused-steps rates, record sweep cost."""
import sys, os, time, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch_cfd_amd as tc
from torch_cfd_amd.initial_conditions import vorticity_field
dev = torch.device('cuda'); torch.set_default_dtype(torch.float64)
n, B, L = int(os.environ.get("N", 512)), int(os.environ.get("B", 64)), 2 * math.pi
grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.0, smooth=True, solver=tc.RK4CrankNicolsonStepper()).to(dev)
def t(fn, reps=1):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
with torch.no_grad():
    w = tc.fft_plan(n, torch.complex128, dev).rfft2(torch.cat([vorticity_field(grid, 4, batch_seeds=list(range(i, i + 16)), device=dev) for i in range(0, B, 16)]))
    K = 50
    print("fused   %.3f ms/step" % (t(lambda: op(w, 1e-3, steps=K)) / K * 1e3))
    def percall():
        x = w
        for _ in range(K): x, _ = op(x, 1e-3)
    print("percall %.3f ms/step" % (t(percall) / K * 1e3))
    out, dw = op(w, 1e-3)
    print("record sweep %.3f ms" % (t(lambda: op.stream_and_residual(out, dw), 5) * 1e3))
    print("explicit     %.3f ms" % (t(lambda: op.explicit_terms(w), 5) * 1e3))
