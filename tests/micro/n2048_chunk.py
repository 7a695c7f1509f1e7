"""2048^2 x 16 fp64 per-call steps/s with the default (unchunked: one field's working set exceeds the Infinity Cache) and with
forced chunks of 1 / 2 fields (TCFD_CHUNK)."""
import json, math, os, subprocess, sys
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import torch_cfd_amd as tc
    from torch_cfd_amd.initial_conditions import vorticity_field
    dev = torch.device("cuda:0")
    torch.set_default_dtype(torch.float64)
    n, B, L = int(os.environ.get('N', 2048)), int(os.environ.get('B', 16)), 2 * math.pi
    grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
    op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.1, forcing_fn=tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=4),
                                   solver=tc.RK4CrankNicolsonStepper()).to(dev)
    w = tc.fft_plan(n, torch.complex128, dev).rfft2(vorticity_field(grid, 4, batch_seeds=list(range(B)), device=dev))
    with torch.no_grad():
        for _ in range(2): w, _ = op(w, 1e-4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6): w, _ = op(w, 1e-4)
        e1.record(); torch.cuda.synchronize()
    print(json.dumps({"n": n, "B": B, "chunk": os.environ.get("TCFD_CHUNK", "default"), "ms_per_step": round(e0.elapsed_time(e1) / 6, 3),
                      "finite": bool(torch.isfinite(w.real).all())}))
else:
    for c in ("", "1", "2", "4"):
        env = dict(os.environ)
        if c: env["TCFD_CHUNK"] = c
        else: env.pop("TCFD_CHUNK", None)
        print(subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
