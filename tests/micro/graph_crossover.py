"""steps/s of forward(w, dt, steps=K) with the captured-graph replay of interior steps on (TCFD_GRAPH=1) and off (=0) over small
problem sizes: where does replay stop paying?  (round 3's rule: state <= 16 MB -> graph)"""
import json, math, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(64, 1, "f64"), (128, 1, "f64"), (128, 4, "f32"), (128, 16, "f32"), (256, 1, "f32"), (256, 4, "f32"), (256, 16, "f32"), (256, 16, "f64"),
         (512, 2, "f64"), (512, 8, "f32")]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    import torch_cfd_amd as tc
    from torch_cfd_amd.initial_conditions import vorticity_field
    dev = torch.device("cuda")
    out = {}
    for n, B, tag in CASES:
        real = torch.float64 if tag == "f64" else torch.float32
        torch.set_default_dtype(real)
        L = 2 * math.pi
        grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
        op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.0, smooth=True, solver=tc.RK4CrankNicolsonStepper()).to(dev)
        cdt = torch.complex128 if tag == "f64" else torch.complex64
        with torch.no_grad():
            w = tc.fft_plan(n, cdt, dev).rfft2(vorticity_field(grid, 4, batch_seeds=list(range(B)), device=dev))
            K = 400
            w = op(w, 1e-3, steps=K)[0]; torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t = time.perf_counter(); w = op(w, 1e-3, steps=K)[0]; torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        out[f"{n}x{B}_{tag}"] = round(best / K * 1e6, 2)
    print(json.dumps(out))
else:
    res = {}
    for g in ("1", "0"):
        env = dict(os.environ, TCFD_GRAPH=g)
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        res["graph" if g == "1" else "plain"] = json.loads(o)
    print("case                us/step graph   plain   state KB")
    for n, B, tag in CASES:
        k = f"{n}x{B}_{tag}"
        kb = B * n * (n // 2 + 1) * (16 if tag == "f64" else 8) / 1024
        print(f"{k:18s} {res['graph'][k]:12.2f} {res['plain'][k]:8.2f} {kb:10.0f}")
