"""Strong-scaling proxy on ONE GPU: per-call time of forward(w, dt) at 1024^2 fp64 for B = 64 / N fields (N = 1, 2, 4, 8) --
exact for this path (no in-step communication) -- plus the per-launch durations of a B-field call (library events) so the
gap between the sum of kernel times and the wall time per call is visible."""
import ctypes, json, math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch_cfd_amd as tc
from torch_cfd_amd.initial_conditions import vorticity_field
dev = torch.device("cuda:0")
torch.set_default_dtype(torch.float64)
n = int(os.environ.get("N", 1024)); L = 2 * math.pi
steps = int(os.environ.get("STEPS", 40))
grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
dt = tc.stable_time_step(dx=L / n, dt=None, max_velocity=5.0, max_courant_number=0.5, viscosity=1e-3)
op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.1, forcing_fn=tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=4),
                               solver=tc.RK4CrankNicolsonStepper()).to(dev)
plan_fft = tc.fft_plan(n, torch.complex128, dev)
lib = tc._lib.load()
res = {}
Bs = [int(b) for b in os.environ.get("BS", "64,32,16,8,4").split(",")]
with torch.no_grad():
    for B in Bs:
        w = plan_fft.rfft2(torch.cat([vorticity_field(grid, 4, batch_seeds=list(range(i, min(i + 8, B))), device=dev)
                                      for i in range(0, B, 8)]))
        for _ in range(5): w, _ = op(w, dt)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(steps): w, _ = op(w, dt)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        # host-side cost of a call alone (enqueue only)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): w, _ = op(w, dt)
        host = (time.perf_counter() - t0) / steps * 1e3
        torch.cuda.synchronize()
        r = {"ms_per_call": round(best, 4), "host_enqueue_ms": round(host, 4)}
        plan = op._plan(w)
        max_rec = 8 * 16 * 64 + 64
        kinds = (ctypes.c_int * max_rec)(); ms = (ctypes.c_float * max_rec)(); cnt = ctypes.c_int(0)
        lib.tcfd_ns2d_profile_begin(plan.handle, max_rec)
        for _ in range(8): w, _ = op(w, dt)
        torch.cuda.synchronize()
        lib.tcfd_ns2d_profile_end(plan.handle, max_rec, ctypes.byref(cnt), kinds, ms)
        per = {}
        for i in range(cnt.value): per.setdefault(kinds[i], []).append(ms[i])
        r["kernel_sum_ms_per_call"] = round(sum(sum(v) for v in per.values()) / 8, 4)
        r["avg_us"] = {k: round(sum(v) / len(v) * 1e3, 1) for k, v in sorted(per.items())}
        r["launches_per_call"] = cnt.value // 8
        res[B] = r
        del w
        torch.cuda.empty_cache()
t64 = res.get(64, {}).get("ms_per_call")
if t64:
    for B in Bs: res[B]["speedup_vs_64"] = round(t64 / res[B]["ms_per_call"], 3)
print(json.dumps(res))
