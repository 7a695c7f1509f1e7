"""A/B of the headline region between two builds of the library ON ONE BOX (153.3 -> 136.7 steps/s between
the round-4 and round-5 driver runs with solver kernels said to be unchanged).

    python tests/micro/r06_solver_ab.py ab/r04 .        # trees to compare: each holds torch_cfd_amd/ with its built .so

The trees are measured in alternation (A B A B ...), one fresh process per measurement, so that drift of the box (clocks,
temperature) lands on both.  Every measurement: C3 (1024^2, batch 64, fp64, Kolmogorov forcing), 20 untimed steps, then 4
regions of 20 steps through K x forward(w, dt), then one instrumented pass (library events per launch) and the HBM probe.
Writes gpurun_out/r06_solver_ab.json."""
import ctypes
import json
import math
import os
import statistics
import subprocess
import sys
import time


def measure(root):
    sys.path.insert(0, os.path.abspath(root))
    import torch

    import torch_cfd_amd as tc
    from torch_cfd_amd.initial_conditions import vorticity_field
    assert os.path.abspath(tc.__file__).startswith(os.path.abspath(root)), tc.__file__

    dev = torch.device("cuda", 0)
    f64 = os.environ.get("AB_DTYPE", "f64") == "f64"
    torch.set_default_dtype(torch.float64 if f64 else torch.float32)
    n, B, L, K = int(os.environ.get("AB_N", 1024)), int(os.environ.get("AB_B", 64)), 2 * math.pi, int(os.environ.get("AB_STEPS", 20))
    grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
    dt = tc.stable_time_step(dx=L / n, dt=None, max_velocity=5.0, max_courant_number=0.5, viscosity=1e-3)
    op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.1, smooth=True, forcing_fn=tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=4),
                                   solver=tc.RK4CrankNicolsonStepper()).to(dev)
    plan_fft = tc.fft_plan(n, torch.complex128 if f64 else torch.complex64, dev)
    with torch.no_grad():
        w = plan_fft.rfft2(torch.cat([vorticity_field(grid, 4, batch_seeds=list(range(i, min(i + 8, B))), device=dev) for i in range(0, B, 8)]))
        for _ in range(20):
            w, _ = op(w, dt)
        regions = []
        for _ in range(4):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(K):
                w, _ = op(w, dt)
            torch.cuda.synchronize(dev)
            regions.append((time.perf_counter() - t0) / K * 1e3)
        lib = tc._lib.load()
        plan = op._plan(w)
        max_rec = K * 16 * 64 + 64
        cnt, kinds, ms = ctypes.c_int(0), (ctypes.c_int * max_rec)(), (ctypes.c_float * max_rec)()
        tc._lib.check(lib.tcfd_ns2d_profile_begin(plan.handle, max_rec), "profile_begin")
        for _ in range(K):
            w, _ = op(w, dt)
        torch.cuda.synchronize(dev)
        tc._lib.check(lib.tcfd_ns2d_profile_end(plan.handle, max_rec, ctypes.byref(cnt), kinds, ms), "profile_end")
    per_kind = {}
    for i in range(min(cnt.value, max_rec)):
        per_kind.setdefault(kinds[i], [0, 0.0])
        per_kind[kinds[i]][0] += 1
        per_kind[kinds[i]][1] += ms[i]
    probe = {}
    nbytes = 1 << 30
    a, b = torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for mode, name, x in ((0, "copy_GBps", 2), (1, "read_GBps", 1), (2, "fill_GBps", 1)):
        t_ms = ctypes.c_float(0)
        tc._lib.check(lib.tcfd_hbm_probe(a.data_ptr(), b.data_ptr(), nbytes, mode, 10, ctypes.byref(t_ms), st), "probe")
        probe[name] = round(x * nbytes / (t_ms.value * 1e-3) / 1e9, 1)
    print(json.dumps({"root": root, "regions_ms_per_step": [round(r, 4) for r in regions],
                      "kernel_ms_per_step": {str(k): round(v[1] / K, 4) for k, v in sorted(per_kind.items())},
                      "kernel_launches_per_step": {str(k): v[0] // K for k, v in sorted(per_kind.items())},
                      "event_sum_ms_per_step": round(sum(v[1] for v in per_kind.values()) / K, 4), "hbm_probe": probe}))


def main():
    trees = sys.argv[1:]
    runs = {t: [] for t in trees}
    for rep in range(int(os.environ.get("AB_REPS", 4))):
        for t in trees:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--measure", t], stdout=subprocess.PIPE, timeout=600)
            assert r.returncode == 0, t
            runs[t].append(json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]))
            print(t, runs[t][-1]["regions_ms_per_step"], file=sys.stderr)
    out = {"what": "C3 headline region, alternating fresh processes on one box; ms per RK4-CN step of 64 fields", "trees": {}}
    for t in trees:
        allr = [x for m in runs[t] for x in m["regions_ms_per_step"]]
        kinds = runs[t][0]["kernel_ms_per_step"].keys()
        out["trees"][t] = {"median_ms_per_step": round(statistics.median(allr), 4), "min": min(allr), "max": max(allr),
                           "steps_per_s_median": round(1e3 / statistics.median(allr), 2), "regions": len(allr),
                           "kernel_ms_per_step_median": {k: round(statistics.median(m["kernel_ms_per_step"].get(k, 0.0) for m in runs[t]), 4) for k in kinds},
                           "kernel_launches_per_step": runs[t][0]["kernel_launches_per_step"],
                           "event_sum_ms_per_step_median": round(statistics.median(m["event_sum_ms_per_step"] for m in runs[t]), 4),
                           "hbm_probe_copy_GBps": [m["hbm_probe"]["copy_GBps"] for m in runs[t]], "runs": runs[t]}
    os.makedirs("gpurun_out", exist_ok=True)
    out["config"] = {k: os.environ.get(k) for k in ("AB_N", "AB_B", "AB_DTYPE", "AB_STEPS")}
    json.dump(out, open(os.environ.get("AB_OUT", "gpurun_out/r06_solver_ab.json"), "w"), indent=1)
    print(json.dumps({t: {k: v for k, v in d.items() if k != "runs"} for t, d in out["trees"].items()}, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--measure":
        measure(sys.argv[2])
    else:
        main()
