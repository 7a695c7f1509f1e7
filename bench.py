#!/usr/bin/env python
"""bench.py -- RK4-CN pseudo-spectral steps/s on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2], "C3" in SURVEY.md 8d): Kolmogorov-forced 2-D
turbulence, 1024^2 grid, batch 64, complex128, nu=1e-3, drag 0.1, forcing
sin(4y), dt = stable_time_step(dx, max_velocity 5) = 6.136e-4, McWilliams random
initial vorticity (seeds 0..63) generated on the device.  One "step" = one
NavierStokes2DSpectral.forward(w, dt) call = one full RK4-CN step of all 64 fields
(5 stages, dw/dt included), input resident in HBM.

N > 1: one process per GPU over RCCL.  Either the driver launches the ranks
(`python -m torch.distributed.run ... bench.py --gpus N`: RANK / WORLD_SIZE come
from the environment) or `python bench.py --gpus N` starts them itself (no
torchrun needed: N children of this script, one device each, rendezvous on
127.0.0.1).  Batch elements are independent trajectories, so there is no
data-path collective; RCCL carries the timing barrier, the max-over-ranks
reduction and (C4 job) the hand-over of records to rank 0.
  --scaling weak   (default) every rank advances its own batch of 64 fields;
                   value = aggregate batch-64 steps/s over all ranks.
  --scaling strong the ONE batch of 64 fields is cut across the ranks (64/N
                   fields per GPU); value = steps/s of that batch (north_star's
                   ">= 7x at 8 GPUs" figure).  A weak-scaling run reports the
                   strong-scaling measurement beside it (`strong_scaling`), so
                   one `--gpus N` run holds both.
`c4_ensemble` is BASELINE configs[3]: the 512-sample McWilliams job (512^2, fp64,
100 + 550 steps, 10 records x 4 fields) cut across the N ranks, with the
un-hidden tail of the gather + D2H split out.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, measured
with HIP events recorded by the library on the launch stream during the timed
steps; `cpu_baseline` times the CPU oracle (a torch-CPU restatement of the
reference's op sequence) on this host.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
KIND_NAMES = {0: "k_cols<MODE_A>", 1: "k_rows_advect", 2: "k_cols<MODE_CA>", 3: "k_cols<MODE_C>+dwdt", 4: "k_dwdt", 5: "other"}
# algorithmic bytes per launch in units of S = B*n*m*sizeof(complex)  (SURVEY 8d pass model)
KIND_ALGO_S = {0: 5.0, 1: 5.0, 2: 9.0, 3: 7.0, 4: 3.0}   # MODE_C of a forward() call also reads w_old and writes dw/dt


def baseline_metric():
    """The metric string of BASELINE.json, verbatim (it ships with the repo)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8"))["metric"]
    except Exception:
        return "RK4-CN spectral steps/s at 1024\u00b2 batch64; achieved HBM GB/s vs peak"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=64, help="fields per GPU")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-sfno", action="store_true", help="skip the secondary SFNO config-5 measurement")
    ap.add_argument("--no-probe", action="store_true", help="skip the STREAM-style HBM probe (keeps profiles clean)")
    ap.add_argument("--fused-steps", action="store_true",
                    help="advance all K steps in ONE forward(steps=K) call (amortises the per-call prologue)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch fields per GPU; strong: --batch fields in total, cut across the GPUs")
    ap.add_argument("--no-c4", action="store_true", help="skip the C4 ensemble job (512 samples of 512^2 over the ranks)")
    ap.add_argument("--c4-samples", type=int, default=512)
    ap.add_argument("--c4-only", action="store_true",
                    help="run ONLY the C4 ensemble job and print its JSON (what a multi-rank run starts as a separate job)")
    ap.add_argument("--c4-timeout", type=float, default=180.0, help="seconds the separate C4 job of a multi-rank run may take")
    ap.add_argument("--host-only", action="store_true",
                    help="launcher / timing / reduction path only, on CPU over gloo with a no-op step (no kernels): "
                         "what the CPU test of the N-rank spawner runs")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script on this node, one device each.
    Rank 0 inherits stdout (the ONE JSON line); every other rank's stdout goes to stderr.  Returns the exit status."""
    import socket
    import subprocess

    n = args.gpus
    if not args.host_only:
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} HIP device(s) are visible", file=sys.stderr)
            return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL peer access between the ranks
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    status = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and status == 0:
                status = rc
                print(f"bench.py: rank {r} exited with status {rc}; stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()      # exactly the children started above
        time.sleep(0.05)
    return status


def host_cpu():
    """(model string, physical cores, logical cpus) of this host from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key == "model name":
                model = val
            elif key == "physical id":
                phys = val
            elif key == "core id":
                core = val
            elif not key and phys is not None:   # blank line closes one logical cpu
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(cores) if cores else logical), logical


def cpu_baseline(n, real, dt, seconds):
    """CPU oracle on the host cores: the same workload at B=2 and at B=8, each with the thread count that is
    fastest for it on this host, extrapolated linearly to B=64.  `value` is the better of the two."""
    from oracle import ns2d as O

    L = 2 * math.pi
    t = O.make_tables(n, L, 1e-3, 0.1, True, None, real)
    t.forcing_hat = O.kolmogorov_forcing_hat(n, L, t.kx, t.ky, 1.0, 4, real=real)
    model, physical, logical = host_cpu()

    def run(Bs, budget):
        w = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, real)) for s in range(Bs)])
        with torch.no_grad():
            # pick the thread count that is fastest on this host (all cores is NOT: a 256-thread MKL/OpenMP team on
            # 1024^2 x 2 fields is ~100x slower than 16-32 threads)
            cands = sorted({c for c in (8, 16, 32, 64) if c <= (os.cpu_count() or 1)} or {1})
            best = None
            for c in cands:
                torch.set_num_threads(c)
                O.advance(w, dt, t)  # warm-up (MKL plans, thread team)
                t1 = time.perf_counter()
                O.advance(w, dt, t)
                el1 = time.perf_counter() - t1
                if best is None or el1 < best[1]:
                    best = (c, el1)
                if el1 > budget / 2:
                    break
            torch.set_num_threads(best[0])
            t0 = time.perf_counter()
            steps = 0
            while True:
                w, _ = O.advance(w, dt, t)
                steps += 1
                el = time.perf_counter() - t0
                if el > budget or steps >= 200:
                    break
        per_step = el / steps
        return {"batch": Bs, "threads": best[0], "steps": steps, "seconds": round(el, 2), "ms_per_step": round(per_step * 1e3, 1),
                "steps_per_s_at_B64": 1.0 / (per_step * 64 / Bs)}

    runs = [run(2, seconds * 0.4), run(8, seconds * 0.6)]
    best = max(runs, key=lambda r: r["steps_per_s_at_B64"])
    return {
        "value": best["steps_per_s_at_B64"],
        "unit": "steps/s (batch 64)",
        "cores": physical,                      # physical cores of the host
        "threads": best["threads"],             # threads the better run used: the fastest of a small sweep (more is slower here)
        "logical_cpus": logical,
        "cpu_model": model,
        "kind": "port",
        "runs": runs,
        "sample": f"oracle/ns2d.py (torch-CPU restatement of the reference op sequence), {n}^2 {str(real)[6:]}, timed at B=2 and B=8 "
                  f"(thread count swept per batch size), each extrapolated linearly to B=64; value = the better one (B={best['batch']}, "
                  f"{best['steps']} steps in {best['seconds']}s, {best['ms_per_step']:.0f} ms/step)",
    }


def _cpu_time(fn, budget_s, cands=(8, 16, 32), max_reps=50):
    """Seconds per call of `fn` on the host cores: the thread count that is fastest on this host out of `cands` (one warm-up +
    one timed call each; all 256 logical cpus are ~100x slower than 16-32 for these sizes), then as many calls as fit
    `budget_s` (at least one).  Returns (seconds per call, threads, calls timed)."""
    cands = sorted({c for c in cands if c <= (os.cpu_count() or 1)} or {1})
    best = None
    t_begin = time.perf_counter()
    for c in cands:
        torch.set_num_threads(c)
        fn()                                      # warm-up: MKL plans, thread team
        t0 = time.perf_counter()
        fn()
        el = time.perf_counter() - t0
        if best is None or el < best[1]:
            best = (c, el)
        if time.perf_counter() - t_begin > budget_s * 0.6:
            break
    torch.set_num_threads(best[0])
    left = budget_s - (time.perf_counter() - t_begin)
    reps = int(max(1, min(max_reps, left / max(best[1], 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    per = (time.perf_counter() - t0) / reps
    return min(per, best[1]), best[0], reps


def cpu_baseline_solver_config(n, B_timed, B_config, real, dt, forced, drag, budget_s, steps_per_call=1):
    """`cpu_baseline` object for one solver config: oracle/ns2d.py on the host cores at
    `B_timed` fields, extrapolated linearly to the config's `B_config` when they differ (stated in `sample`)."""
    from oracle import ns2d as O

    L = 2 * math.pi
    t = O.make_tables(n, L, 1e-3, drag, True, None, real)
    if forced:
        t.forcing_hat = O.kolmogorov_forcing_hat(n, L, t.kx, t.ky, 1.0, 4, real=real)
    w = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, real)) for s in range(B_timed)])
    model, physical, logical = host_cpu()
    with torch.no_grad():
        per, threads, reps = _cpu_time(lambda: O.advance(w, dt, t, steps=steps_per_call), budget_s)
    per_step = per / steps_per_call * (B_config / B_timed)
    return {"value": 1.0 / per_step, "unit": f"steps/s (batch {B_config})", "cores": physical, "threads": threads,
            "logical_cpus": logical, "cpu_model": model, "kind": "port",
            "sample": f"oracle/ns2d.py, {n}^2 {str(real)[6:]}, {'Kolmogorov-forced' if forced else 'unforced'}, B={B_timed}, "
                      f"{reps} x {steps_per_call} step(s) at {per / steps_per_call * 1e3:.1f} ms/step with {threads} threads"
                      + (f", extrapolated linearly to B={B_config} (x{B_config // B_timed})" if B_config != B_timed else "")}


def cpu_baseline_sfno(budget_s):
    """`cpu_baseline` for config 5: oracle/sfno.py forward + oracle/fno.py sobolev_loss at b = 4 (the config's b = 32 is 8 x
    that: samples do not interact), same model / input construction as `sfno_config5`."""
    from oracle import fno as OF
    from oracle import sfno as OS
    from torch_cfd_amd import fno

    torch.set_default_dtype(torch.float32)
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in fno.SFNO(24, 24, 5, width=10, num_spectral_layers=4).state_dict().items()}
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(4, 256, 256, 10, generator=g)
    y = torch.randn(4, 256, 256, 10, generator=g)

    def fwd_loss():
        out = OS.sfno_forward(sd, x, (24, 24, 5), width=10, num_hidden=3, out_steps=10)
        return OF.sobolev_loss(out, y, 256, norm_order=0, relative=True)

    model, physical, logical = host_cpu()
    with torch.no_grad():
        per, threads, reps = _cpu_time(fwd_loss, budget_s, cands=(16, 32), max_reps=3)
    return {"value": 4 / per, "unit": "samples/s (forward + loss)", "cores": physical, "threads": threads, "logical_cpus": logical,
            "cpu_model": model, "kind": "port",
            "sample": f"oracle/sfno.py + oracle/fno.py sobolev_loss, b=4 of the config's 32 (samples are independent: x8 stated, "
                      f"samples/s unchanged), {reps} call(s) at {per * 1e3:.0f} ms with {threads} threads"}


FNO_KINDS = {0: "fwd_ty", 1: "fwd_x", 2: "contract", 3: "inv_x", 4: "inv_ty", 5: "pointwise", 6: "pointwise_bwd", 7: "pointwise_1layer",
             8: "contract_wgrad", 9: "other", 10: "pointwise_1layer_bwd"}


def fno_kernel_times(fn, dev, reps=3, cap=4096):
    """Per-kernel-kind launch durations of the FNO library while `fn` runs `reps` times: HIP events recorded by the library
    around every launch on its launch stream (tcfd_fno_profile_begin / _end) -- i.e. the kernels as they run INSIDE the model,
    behind each other's cache state, which is what a rocprofv3 kernel trace of the same command sees.
    Returns {kind: {"launches": per call of fn, "avg_ms": ..., "total_ms": per call}}."""
    import ctypes

    from torch_cfd_amd import _lib
    lib = _lib.load()
    fn(); torch.cuda.synchronize(dev)
    _lib.check(lib.tcfd_fno_profile_begin(cap), "tcfd_fno_profile_begin")
    try:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
    finally:
        count = ctypes.c_int(0)
        kinds = (ctypes.c_int * cap)()
        ms = (ctypes.c_float * cap)()
        _lib.check(lib.tcfd_fno_profile_end(cap, ctypes.byref(count), kinds, ms), "tcfd_fno_profile_end")
    out = {}
    for i in range(min(count.value, cap)):
        d = out.setdefault(FNO_KINDS.get(kinds[i], str(kinds[i])), {"n": 0, "t": 0.0, "max": 0.0})
        d["n"] += 1; d["t"] += ms[i]; d["max"] = max(d["max"], ms[i])
    return {k: {"launches": d["n"] // reps, "avg_ms": round(d["t"] / d["n"], 4), "max_ms": round(d["max"], 4),
                "total_ms": round(d["t"] / reps, 4)} for k, d in out.items()}


def sfno_width_line(dev, width, b=32, act="ReLU", steps=3):
    """One line per model width ("additionally report width 32"; 16 / 20 are the reference's other widths,
    fno/sfno_pytest.py:261 and its notebooks): SFNO(24,24,5,width) on (b,256,256,10) -- forward, forward + loss, one training step
    (median), on the 21.5 A_H byte model of the width-10 line, and the in-model duration of the pointwise backward kernel."""
    from torch_cfd_amd import fno

    torch.set_default_dtype(torch.float32)
    torch.manual_seed(0)
    model = fno.SFNO(24, 24, 5, width=width, num_spectral_layers=4, activation=act).to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(b, 256, 256, 10, generator=g).to(dev)
    y = torch.randn(b, 256, 256, 10, generator=g).to(dev)
    loss_fn = fno.SobolevLoss(n_grid=256, norm_order=0, relative=True).to(dev)

    def timeit(fn, n):
        fn(); torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n

    with torch.no_grad():
        t_fwd = timeit(lambda: model(x), 5)
        t_all = timeit(lambda: loss_fn(model(x), y), 5)
    model.train()

    def train_step():
        model.zero_grad(set_to_none=True)
        loss_fn(model(x), y).backward()

    train_step(); torch.cuda.synchronize(dev)
    per = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); train_step(); e1.record(); torch.cuda.synchronize(dev)
        per.append(e0.elapsed_time(e1))
    t_train = sorted(per)[len(per) // 2]
    kt = fno_kernel_times(train_step, dev, reps=2)
    fell_back = []
    saved = fno._pointwise_reference

    def spy(*a, **k):
        fell_back.append(1)
        return saved(*a, **k)
    fno._pointwise_reference = spy
    try:
        train_step(); torch.cuda.synchronize(dev)
    finally:
        fno._pointwise_reference = saved
    peak_gb = torch.cuda.max_memory_allocated(dev) / 1e9
    A_H = b * width * 256 * 256 * 10 * 4
    algo_gb = 21.5 * A_H / 1e9
    # MACs per point of the forward's pointwise blocks (4 hidden-layer blocks incl. the lifting tail): W1 + W2 (+ Ws)
    pw_flop = 2.0 * b * 256 * 256 * 10 * (4 * 2 * 4 * width * width + 3 * width * width)
    del model, x, y
    torch.cuda.empty_cache()
    return {"workload": f"SFNO(24,24,5,width={width},layers=4,{act}) on x ({b},256,256,10) fp32, synthetic",
            "forward_ms": round(t_fwd, 3), "forward_plus_loss_ms": round(t_all, 3), "train_step_ms": round(t_train, 2),
            "algo_GB": round(algo_gb, 2), "algo_GBps": round(algo_gb / (t_all * 1e-3), 1),
            "frac_of_hbm_peak": round(algo_gb / (t_all * 1e-3) / HBM_PEAK_GBS, 4),
            "pointwise_fp32_TFLOPs_in_forward": round(pw_flop / (t_fwd * 1e-3) / 1e12, 1),
            "also_compute_bound": f"the forward's pointwise blocks alone are {pw_flop / 1e12:.3f} TFLOP = "
                                  f"{pw_flop / 157.3e12 * 1e3:.2f} ms at the 157.3 TFLOP/s fp32 peak (vector = matrix rate on gfx950)",
            "train_step_ms_per_GB": round(t_train / algo_gb, 3), "einsum_recompute_fallbacks_in_a_training_step": len(fell_back),
            "train_kernels": {k: v for k, v in kt.items() if k in ("pointwise_bwd", "pointwise", "inv_ty", "fwd_ty")},
            "peak_memory_GB": round(peak_gb, 1)}


def sfno_config5(dev, with_cpu=True):
    """Secondary measurement (BASELINE configs[4], SURVEY 8d "C5"): SFNO(24,24,5, width 10, 4 layers) forward +
    SobolevLoss on x = randn(32,256,256,10) fp32, random-init weights (seed 0); plus one training step
    (forward + loss + backward).  Algorithmic bytes: 21.5 A_H = 18.0 GB per forward+loss."""
    from torch_cfd_amd import fno

    torch.set_default_dtype(torch.float32)
    torch.manual_seed(0)
    model = fno.SFNO(24, 24, 5, width=10, num_spectral_layers=4).to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(32, 256, 256, 10, generator=g).to(dev)
    y = torch.randn(32, 256, 256, 10, generator=g).to(dev)
    loss_fn = fno.SobolevLoss(n_grid=256, norm_order=0, relative=True).to(dev)

    def timeit(fn, n):
        fn(); torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n

    with torch.no_grad():
        t_fwd = timeit(lambda: model(x), 10)
        t_all = timeit(lambda: loss_fn(model(x), y), 10)
        # the same model with the activation of the reference's training script (fno/train.py:303 --activation GELU; the class
        # default, which BASELINE configs[4] is quoted on, is ReLU): exact GELU as a packed branch-free 2^-s(|v|) evaluation
        gelu_model = fno.SFNO(24, 24, 5, width=10, num_spectral_layers=4, activation="GELU").to(dev).eval()
        t_gelu = timeit(lambda: gelu_model(x), 10)
    # ... and its training step (GELU keeps the block's pre-activation for the backward: tcfd_fno_pointwise_pre)
    gelu_model.train()

    def gelu_train_step():
        gelu_model.zero_grad(set_to_none=True)
        loss_fn(gelu_model(x), y).backward()
    gelu_train_step(); gelu_train_step(); gelu_train_step(); torch.cuda.synchronize(dev)
    gper = []
    for _ in range(5):          # median of 5, as for the ReLU step below (a step that meets an allocator round trip is 2 ms slower)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gelu_train_step(); e1.record(); torch.cuda.synchronize(dev)
        gper.append(e0.elapsed_time(e1))
    t_train_gelu = sorted(gper)[2]
    gelu_kernels = fno_kernel_times(gelu_train_step, dev, reps=2)
    del gelu_model
    torch.cuda.empty_cache()
    model.train()

    def train_step():
        model.zero_grad(set_to_none=True)
        loss_fn(model(x), y).backward()

    # every step timed on its own, median of 5: a training step allocates ~11 GB through the caching allocator, and one
    # slow step (an allocator round trip to the driver) used to move a 3-step mean by 10 ms from run to run
    train_step(); train_step(); torch.cuda.synchronize(dev)
    per_step = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); train_step(); e1.record(); torch.cuda.synchronize(dev)
        per_step.append(e0.elapsed_time(e1))
    t_train = sorted(per_step)[len(per_step) // 2]
    train_kernels = fno_kernel_times(train_step, dev, reps=2)
    model.eval()
    with torch.no_grad():
        fwd_kernels = fno_kernel_times(lambda: model(x), dev, reps=3)
    model.train()
    algo_gb = 21.5 * 32 * 10 * 256 * 256 * 10 * 4 / 1e9
    # roofline of the dominant kernel of the forward (k_pointwise<10,40,10>: FFN + skip conv + activation of a hidden layer
    # in one pass, 4 launches = 2.3 of the 5.4 ms): it reads the spectral-conv output and the layer input and writes the
    # next activation -- 3 A_H of algorithmic bytes per launch; timed in isolation with events on torch's
    # current stream, which is the stream the library launches on
    roof = None
    try:
        A_H = 32 * 10 * 256 * 256 * 10 * 4
        x1, v = torch.randn(32, 10, 256, 256, 10, device=dev), torch.randn(32, 10, 256, 256, 10, device=dev)
        mlp, w, act = model.mlp[0], model.w[0], model.activations[0]
        with torch.no_grad():
            blk = lambda: fno.hip_pointwise(x1, mlp.linear1, mlp.activation, mlp.linear2, skip=v, skip_conv=w, act2=act)
            assert blk() is not None
            t_blk = timeit(blk, 20)
            # the backward of the same block (k_pwb_tiles, csrc/tcfd_fno_tiles.hip): reads x, skip, dout and the block's forward output
            # (its ReLU mask, tcfd_fno_pointwise_bwd_out), writes dx, dskip = 6 A_H
            spec = (True, mlp.activation, act, 1, None)
            y_blk = blk()
            keeps = fno._saved_kind(spec, 10, 40, 10, 256 * 256 * 10) == 1    # ReLU / ReLU (the reference's default): the output is handed over
            bwd = lambda: fno._hip_pointwise_backward(spec, x1, v, v, mlp.linear1.weight, mlp.linear1.bias, mlp.linear2.weight,
                                                      mlp.linear2.bias, w.weight, w.bias, None, None, out=y_blk if keeps else None)
            t_bwd = timeit(bwd, 10)
            del y_blk
            n_mfma = 59                              # v_mfma_f32_16x16x4_f32 per 16 points at width 10 (header of tcfd_fno_tiles.hip)
            useful_mac = 2260                        # W1 x, W2^T g2, W1^T g1, Ws^T g2, three weight-gradient outer products, biases
        del x1, v
        # `roofline` prices the kernel AS IT RUNS INSIDE model(x) (library events around each launch, fno_kernel_times): that is
        # what profiles/r05_sfno_forward_kernel_stats.csv shows.  The isolated loop on fresh randn tensors above reads slower
        # (every launch finds its two 839 MB inputs cold in every cache; in the model the skip input was written one kernel
        # earlier and part of it is still in the 256 MB Infinity Cache) and is reported beside it as `isolated_launch_ms`.
        pw_in_model = fwd_kernels.get("pointwise", {})
        t_iso, t_bwd_iso = t_blk, t_bwd
        if pw_in_model.get("launches") == 4:
            t_blk = pw_in_model["avg_ms"]
        bwd_in_model = train_kernels.get("pointwise_bwd", {})
        if bwd_in_model.get("launches") == 4:
            t_bwd = bwd_in_model["avg_ms"]
        ach = 3 * A_H / (t_blk * 1e-3) / 1e9
        # L2 <-> memory bytes per launch from the rocprofv3 PMC passes of tests/prof_sfno.sh (profiles/sfno_traffic.json ships with
        # the repo, it is not re-measured by this run): (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the guide's gfx950 correction
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "sfno_traffic.json")))
        except Exception:
            tj = {}
        pw_key = next((k for k in tj if k.startswith("k_pointwise<10, 40, 10")), None)
        kern_table = {k: {kk: vv for kk, vv in v.items() if kk in ("launches_in_profile", "avg_us", "algo_bytes", "algo_TBps", "traffic_bytes",
                                                                    "l2_hit", "lds_conflict_share", "what")}
                      for k, v in tj.items() if isinstance(v, dict)}
        roof = {"kernel": "k_pointwise<10,40,10> (FFN + skip conv + activation of one hidden layer)", "bound": "hbm",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "algo_bytes_per_launch": 3 * A_H, "avg_launch_ms": round(t_blk, 4), "launches_per_forward": 4,
                "timed": "HIP events recorded by the library around each launch inside model(x) (tcfd_fno_profile_begin/_end), 3 forwards",
                "isolated_launch_ms": round(t_iso, 4), "kernels_in_model": fwd_kernels,
                "traffic": tj.get(pw_key, {}).get("traffic_bytes") if pw_key else None,
                "traffic_source": ("profiles/sfno_traffic.json (rocprofv3 --pmc passes of tests/bench_sfno.py, tests/prof_sfno.sh): "
                                   "bytes between L2 and the memory side per launch; the 839 MB activations exceed the Infinity Cache"
                                   ) if pw_key else None,
                "vector_work": "900 FMAs per point as v_pk_fma_f32 = 0.24 ms per launch at the 157 TFLOP/s packed-fp32 peak (~0.3 ms of "
                               "VALU issue with the ReLUs and address arithmetic), beside 0.31 ms of HBM time at 8 TB/s.  Until round 4 the "
                               "run-time activation switch inside the hidden-unit loop added 1,224 scalar instructions and ~10 taken "
                               "branches per wave (SQ_INSTS_SALU ~ SQ_INSTS_VALU in profiles/r04_sfno_pmc.txt): 571 us; with the "
                               "activations as template parameters 475 us; with the two inputs read non-temporally (late round 4) 444 us",
                "kernels_from_profile": kern_table or None,
                "backward_kernel": {"kernel": "k_pwb_tiles<10,40,10> (tiled all-MFMA backward of the block; in-model launch time)",
                                    "algo_bytes_per_launch": (6 if keeps else 5) * A_H, "avg_launch_ms": round(t_bwd, 4),
                                    "timed": "library events around each launch inside the training step, 2 steps",
                                    "isolated_launch_ms": round(t_bwd_iso, 4), "kernels_in_training_step": train_kernels,
                                    "achieved": round((6 if keeps else 5) * A_H / (t_bwd * 1e-3) / 1e9, 1),
                                    "frac": round((6 if keeps else 5) * A_H / (t_bwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "mfma": {"bound": "mfma", "unit": "TFLOP/s", "peak": 157.3,
                                             # issued: v_mfma_f32_16x16x4_f32 (2048 flop each) per 16 points, tile padding included
                                             "mfma_per_16_points": n_mfma,
                                             "issued_flop_per_launch": n_mfma * 2048 * (A_H // 40 // 16),
                                             "achieved": round(n_mfma * 2048 * (A_H // 40 // 16) / (t_bwd * 1e-3) / 1e12, 1),
                                             "frac": round(n_mfma * 2048 * (A_H // 40 // 16) / (t_bwd * 1e-3) / 1e12 / 157.3, 4),
                                             # useful: the multiply-adds the block's backward needs per point -- W1 x (for the mask and
                                             # h: 400), W2^T g2 (400), W1^T g1 (400), Ws^T g2 (100), the three outer products (400 + 400 + 100)
                                             # and the bias sums (60); nothing of z2 is recomputed (the output's sign is the mask).  The rest
                                             # of the issued work is the zero padding of width 10 / 40 into 16 x 16 x 4 tiles.
                                             "useful_flop_per_launch": 2 * useful_mac * (A_H // 40),
                                             "useful_achieved": round(2 * useful_mac * (A_H // 40) / (t_bwd * 1e-3) / 1e12, 1),
                                             "useful_frac": round(2 * useful_mac * (A_H // 40) / (t_bwd * 1e-3) / 1e12 / 157.3, 4),
                                             "useful_over_issued": round(2 * useful_mac * 16 / (n_mfma * 2048), 3)},
                                    "note": f"{n_mfma} v_mfma_f32_16x16x4_f32 per 16 points (round 3: 105 -> 93 by the tile row map; round 4: 71 -- output "
                                            "mask from the forward output, transpositions as products with the identity; round 5: 59 -- the "
                                            "transpositions through wave-private LDS, every tensor read once as 16-byte lanes, weights as LDS "
                                            "fragments: one kernel for every width 4 ... 32); peak = dense fp32 matrix rate, 256 flop/clk/CU x 256 CUs x 2.4 GHz"}}
    except Exception as e:
        roof = {"error": repr(e)}
    base = None
    if with_cpu:
        try:
            base = cpu_baseline_sfno(14.0)
        except Exception as e:
            base = {"value": None, "error": repr(e)}
    return {"roofline": roof, "cpu_baseline": base,
            "gpu_over_cpu": round(32 / (t_all * 1e-3) / base["value"], 1) if base and base.get("value") else None,
            "workload": "SFNO(24,24,5,width=10,layers=4) forward + SobolevLoss, x (32,256,256,10) fp32, synthetic",
            "forward_ms": round(t_fwd, 3), "forward_plus_loss_ms": round(t_all, 3), "train_step_ms": round(t_train, 2),
            "forward_ms_with_gelu": round(t_gelu, 3), "train_step_ms_gelu": round(t_train_gelu, 2),
            "train_step_ms_gelu_each": [round(t, 2) for t in gper],
            "gelu_training_kernels": {k: v for k, v in gelu_kernels.items() if k in ("pointwise_bwd", "pointwise")},
            "train_step_ms_each": [round(t, 2) for t in per_step],
            "samples_per_s": round(32 / (t_all * 1e-3), 1), "algo_GB": round(algo_gb, 2),
            "algo_GBps": round(algo_gb / (t_all * 1e-3), 1), "frac_of_hbm_peak": round(algo_gb / (t_all * 1e-3) / HBM_PEAK_GBS, 4)}


def sfno_notebook_training(dev):
    """The one training-loop figure the reference prints for this path (not the headline metric):
    examples/ex2_SFNO_train.ipynb -- SFNO(32, 32, 5, width 10), batch 4, 64 x 64 x 10 -> 10 steps, Adam, SobolevLoss(order 0,
    relative) -- 33-39 it/s on an unnamed GPU.  Same model / optimiser / loss on synthetic data of that shape: it/s of
    zero_grad + forward + loss + backward + optimiser step."""
    from torch_cfd_amd import fno

    torch.set_default_dtype(torch.float32)
    torch.manual_seed(0)
    model = fno.SFNO(32, 32, 5, 10, beta=-1e-2).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    loss_fn = fno.SobolevLoss(n_grid=64, norm_order=0, time_average=True, relative=True).to(dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(4, 64, 64, 10, generator=g).to(dev)
    y = torch.randn(4, 64, 64, 10, generator=g).to(dev)

    def it():
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(model(x), y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        it()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    n_it = 50
    for _ in range(n_it):
        last = it()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    graphed = None
    try:     # the same iteration captured once and replayed (fno.make_graphed_training_step): the eager loop is host-bound
        torch.manual_seed(0)
        model_g = fno.SFNO(32, 32, 5, 10, beta=-1e-2).to(dev).train()
        opt_g = torch.optim.Adam(model_g.parameters(), lr=1e-3, capturable=True)
        step = fno.make_graphed_training_step(model_g, loss_fn, opt_g, x, y)
        for _ in range(3):
            step(x, y)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(n_it):
            lg = step(x, y)
        torch.cuda.synchronize(dev)
        elg = time.perf_counter() - t1
        graphed = {"iterations_per_s": round(n_it / elg, 1), "ms_per_iteration": round(elg / n_it * 1e3, 3), "finite": bool(torch.isfinite(lg))}
    except Exception as e:
        graphed = {"error": repr(e)}
    return {"workload": "SFNO(32,32,5,width=10) training loop of examples/ex2_SFNO_train.ipynb: batch 4, 64x64x10, Adam, SobolevLoss "
                        "(order 0, relative); synthetic data",
            "iterations_per_s": round(n_it / el, 1), "ms_per_iteration": round(el / n_it * 1e3, 3), "finite": bool(torch.isfinite(last)),
            "graph_replay": graphed,
            "reference_printed": "33-39 it/s on an unnamed GPU (examples/ex2_SFNO_train.ipynb:147-371; other hardware, not comparable)"}


def other_baseline_configs(dev, with_cpu=True):
    """Secondary lines for the other single-GPU BASELINE configs on the same kernels: C2 = 256^2, B=16,
    fp32, unforced McWilliams, dt=1e-3 (1000-step job, measured over 400 steps through forward(w, dt, steps=k)); C4 per-GPU
    shard = 512^2, B=64, fp64, unforced, dt=1e-3; plus two sizes outside BASELINE: 768^2 (n = 3 * 2^k on the fused kernels)
    and 2048^2 (one field nearly fills the Infinity Cache: single-field chunks).  value = batch steps/s."""
    import torch_cfd_amd as tc
    from torch_cfd_amd.initial_conditions import vorticity_field

    out = {}
    # (name, n, B, dtype, steps, one forward(steps=k) call?, Kolmogorov forcing + drag 0.1?)
    for name, n, B, real, steps, fused, forced in (("C1_128x1_f64", 128, 1, torch.float64, 200, True, True),
                                                   ("C2_256x16_f32", 256, 16, torch.float32, 400, True, False),
                                                   ("C4_shard_512x64_f64", 512, 64, torch.float64, 40, False, False),
                                                   ("C4_shard_512x64_f32", 512, 64, torch.float32, 40, False, False),   # "also report c64"
                                                   ("n768x64_f64", 768, 64, torch.float64, 20, False, False),   # n = 3 * 2^k: radix-12 first pass
                                                   ("n2048x16_f64", 2048, 16, torch.float64, 6, False, False)):  # one field per chunk
        torch.set_default_dtype(real)
        L = 2 * math.pi
        grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
        forcing = tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=4) if forced else None
        op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.1 if forced else 0.0, smooth=True, forcing_fn=forcing,
                                       solver=tc.RK4CrankNicolsonStepper()).to(dev)
        cdt = torch.complex64 if real == torch.float32 else torch.complex128
        with torch.no_grad():
            w = tc.fft_plan(n, cdt, dev).rfft2(torch.cat([vorticity_field(grid, 4, batch_seeds=list(range(i, min(i + 8, B))),
                                                                          device=dev) for i in range(0, B, 8)]))
            run = (lambda w: op(w, 1e-3, steps=steps)[0]) if fused else (lambda w: [w := op(w, 1e-3)[0] for _ in range(steps)][-1])
            w = run(w)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            w = run(w)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
        S = B * n * (n // 2 + 1) * (8 if real == torch.float32 else 16)
        del w
        torch.cuda.empty_cache()
        out[name] = {"steps_per_s": round(steps / el, 1), "ms_per_step": round(el / steps * 1e3, 4),
                     "step_algo_GBps": round(70.0 * S * steps / el / 1e9, 1),
                     "api": "forward(w,dt,steps=k)" if fused else "k x forward(w,dt)"}
    if with_cpu:
        # the CPU restatement beside every BASELINE config line, bounded: ~8 s per config
        for name, args_ in (("C1_128x1_f64", (128, 1, 1, torch.float64, 1e-3, True, 0.1, 6.0, 20)),
                            ("C2_256x16_f32", (256, 16, 16, torch.float32, 1e-3, False, 0.0, 8.0, 1)),
                            ("C4_shard_512x64_f64", (512, 8, 64, torch.float64, 1e-3, False, 0.0, 8.0, 1))):
            try:
                base = cpu_baseline_solver_config(*args_)
                out[name]["cpu_baseline"] = base
                out[name]["gpu_over_cpu"] = round(out[name]["steps_per_s"] / base["value"], 1)
            except Exception as e:
                out[name]["cpu_baseline"] = {"value": None, "error": repr(e)}
    return out


def host_only_run(args, world, rank, real_stdout):
    """Launcher check (`--host-only`): process group over gloo on CPU, the barrier / max-reduce / gather-of-rates path
    and the JSON line, around a step that does nothing.  No kernels, no device: it measures nothing about the path."""
    import torch.distributed as dist

    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_cfd_amd.distributed import shard_batch

    lo, hi = shard_batch(args.batch, rank, world) if args.scaling == "strong" else (rank * args.batch, (rank + 1) * args.batch)
    state = torch.zeros(8)
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state = state + 1.0
    if dist.is_initialized():
        dist.barrier()
    el = time.perf_counter() - t0
    rates = [args.steps / el]
    spans = [[lo, hi]]
    if dist.is_initialized():
        mine = torch.tensor([lo, hi], dtype=torch.int64)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)          # what every rank REALLY took (the test compares it with shard_batch)
        spans = [g.tolist() for g in got]
        t = torch.tensor([el], dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rates = [args.steps / max(e.item(), 1e-12) for e in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    out = {"metric": baseline_metric(), "value": None, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 6), "higher_is_better": True,
           "scaling": args.scaling, "vs_baseline": None, "dtype": None, "data": "none (host-only launcher check, no kernels)",
           "dry_run": True, "backend": "gloo", "world_size": dist.get_world_size() if dist.is_initialized() else 1,
           "fields_of_rank0": [lo, hi], "fields_of_every_rank": spans, "per_rank_steps_per_s": [round(r, 1) for r in rates],
           "spawned_by_bench": os.environ.get("BENCH_SPAWNED") == "1"}
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def c4_ensemble(dev, world, rank, total, as_rank0_of=None):
    """BASELINE configs[3] as ONE job cut across the ranks (strong scaling of the data-generation loop of
    fno/data_gen/data_gen_McWilliams2d.py:126-152): `total` McWilliams samples of 512^2 in batches of 64, fp64, 100 warm-up
    + 550 recorded steps, a record every 55 (10 records x 4 fields), c2r + 2x subsample + fp32 cast on the device, records
    handed to rank 0's page-locked host memory WHILE the steps go on (RCCL peers -> rank 0, then PCIe)."""
    import torch.distributed as dist

    from torch_cfd_amd.data_gen import generate_mcwilliams_dataset

    torch.set_default_dtype(torch.float64)
    stats = {}
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    data = generate_mcwilliams_dataset(512, total, 64, 1e-3, 100, 550, 55, viscosity=1e-3, peak_wavenumber=4, random_state=0,
                                       subsample=2, dtype=torch.float32, cdtype=torch.complex64, device=dev, stats=stats,
                                       as_rank0_of=as_rank0_of)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    phases = torch.tensor([el, stats["setup_s"], stats["stepping_s"], stats["handover_tail_s"]], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        dist.all_reduce(phases, op=dist.ReduceOp.MAX)
    el, setup, stepping, tail = phases.tolist()
    if rank != 0:
        return None
    if as_rank0_of is not None:      # the proxy: only rank 0's rows of the result exist
        return {"seconds": round(el, 3), "setup_s": round(setup, 3), "stepping_s": round(stepping, 3),
                "handover_tail_s": round(tail, 3), "samples": stats["samples"]}
    ok = all(bool(torch.isfinite(v).all()) for v in data.values() if v.is_floating_point())
    gb = sum(v.numel() * v.element_size() for v in data.values()) / 1e9
    return {"workload": f"McWilliams ensemble, {total} samples of 512^2 in batches of 64 over {world} GPU(s), fp64, 100 + 550 "
                        "steps, 10 records x 4 fields -> (N, 10, 256, 256) fp32 on the host of rank 0",
            "seconds": round(el, 3), "sample_steps_per_s": round(total * 650 / el, 1),
            "setup_s": round(setup, 3), "stepping_s": round(stepping, 3), "handover_tail_s": round(tail, 3),
            "dataset_GB": round(gb, 3), "finite": ok, "scaling": "strong (fixed 512-sample job)",
            "note": "max over ranks of each phase; handover_tail_s = gather + D2H left after the last step finished"}


LAUNCH_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
              "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "BENCH_SPAWNED", "BENCH_FORCE_DIST", "TORCHELASTIC_RUN_ID",
              "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")


def c4_as_separate_job(world, samples, timeout):
    """The C4 ensemble job of a multi-rank run, started by rank 0 as its OWN set of `world` ranks (this script with --c4-only,
    through the spawner above) with a time limit: its hand-over uses point-to-point RCCL traffic that the step benchmark does
    not, and whatever happens to it -- an error, a hang -- must not take the scaling measurement down with it.  The parent
    ranks idle meanwhile (their GPUs are shared with the job: a few GB of their 288)."""
    import signal
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--c4-only", "--c4-samples", str(samples)]
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, start_new_session=True)
        try:
            out, _ = proc.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)      # the job's own process group (start_new_session): the spawner and its ranks
            proc.communicate()
            return {"error": f"the separate C4 job did not finish within {timeout:.0f} s and was stopped"}
        lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
        if proc.returncode != 0 or not lines:
            return {"error": f"the separate C4 job exited with status {proc.returncode}"}
        return json.loads(lines[-1])
    except Exception as e:
        return {"error": repr(e)}


def c4_only_main(args):
    """`--c4-only`: the C4 job alone, one JSON object on stdout (rank 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        dist.init_process_group("nccl", device_id=dev)
    res = c4_ensemble(dev, world, rank, args.c4_samples)
    if rank == 0:
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args, argv))       # no launcher: this process only starts the ranks and forwards the status
    if args.c4_only:
        return c4_only_main(args)
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio on stdout (flushed at exit,
    # i.e. AFTER anything Python prints): keep a private handle on the real stdout for the JSON line and point fd 1 at
    # stderr for everything else (libraries, warnings, the banner).
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" in os.environ and args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}",
              file=sys.stderr)
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.host_only:
        return host_only_run(args, world, rank, real_stdout)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # the env switch exercises the RCCL path on 1 GPU
    dist = None
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")    # one node: the host-side group must not depend on the hostname resolving
        import datetime

        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=30))
    # host-side barrier for the wait on the separate C4 job (an RCCL barrier would keep a spinning kernel on every waiting GPU)
    host_group = dist.new_group(backend="gloo") if (use_dist and world > 1) else None

    import torch_cfd_amd as tc
    from torch_cfd_amd.distributed import shard_batch
    from torch_cfd_amd.initial_conditions import vorticity_field

    real, cdt = (torch.float64, torch.complex128) if args.dtype == "f64" else (torch.float32, torch.complex64)
    torch.set_default_dtype(real)
    n, L = args.n, 2 * math.pi
    m = n // 2 + 1
    grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
    dt = tc.stable_time_step(dx=L / n, dt=None, max_velocity=5.0, max_courant_number=0.5, viscosity=1e-3)
    forcing = tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=4)
    op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.1, smooth=True, forcing_fn=forcing,
                                   solver=tc.RK4CrankNicolsonStepper()).to(dev)
    plan_fft = tc.fft_plan(n, cdt, dev)
    csize = 16 if real == torch.float64 else 8

    def initial_state(seeds):
        with torch.no_grad():
            if not seeds:
                return torch.empty(0, n, m, dtype=cdt, device=dev)
            return plan_fft.rfft2(torch.cat([vorticity_field(grid, 4, batch_seeds=seeds[i:i + 8], device=dev)
                                             for i in range(0, len(seeds), 8)]))

    def advance(w, k):
        if args.fused_steps:
            out, _ = op(w, dt, steps=k)
            return out
        for _ in range(k):
            w, _ = op(w, dt)
        return w

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(w, warmup, steps):
        """W untimed warm-up steps, then exactly K steps between two fences; (state, this rank's seconds)."""
        with torch.no_grad():
            w = advance(w, warmup)
            fence()
            t0 = time.perf_counter()
            w = advance(w, steps)
            fence()
            return w, time.perf_counter() - t0

    def over_ranks(elapsed):
        """(max over ranks, [per-rank seconds])."""
        if not use_dist:
            return elapsed, [elapsed]
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        every = [e.item() for e in every]
        return max(every), every

    # ---- the two batch layouts.  weak: --batch fields on EVERY rank (seeds rank*B ...); strong: --batch fields in total
    weak_seeds = [rank * args.batch + i for i in range(args.batch)]
    lo, hi = shard_batch(args.batch, rank, world)
    strong_seeds = list(range(lo, hi))
    primary_seeds = weak_seeds if args.scaling == "weak" else strong_seeds
    B = len(primary_seeds)                      # fields this rank advances in the headline measurement
    S = B * n * m * csize

    w = initial_state(primary_seeds)
    plan = op._plan(w)
    lib = tc._lib.load()

    # Two back-to-back regions of the same K steps:
    #   1. the TIMED region (value, ms_per_step): no instrumentation at all;
    #   2. the INSTRUMENTED region: the library brackets every launch with HIP events on the launch stream.
    # They are separate because a chunked step is ~176 short launches and an event pair per launch costs ~7 us of
    # host time and in-order queue barriers each (measured: 7.56 -> 8.78 ms per step) -- bracketing the timed region
    # itself would slow down the very number it annotates.  `instrumented_ms_per_step` shows the difference.
    w, elapsed_local = timed(w, args.warmup, args.steps)
    elapsed, per_rank = over_ranks(elapsed_local)
    cnt = ctypes.c_int(0)
    max_rec = args.steps * 16 * 64 + 64   # every launch of every chunk of every step (<= 64 chunks)
    kinds = (ctypes.c_int * max_rec)()
    ms = (ctypes.c_float * max_rec)()
    elapsed_instr = float("nan")
    if B > 0:
        with torch.no_grad():
            tc._lib.check(lib.tcfd_ns2d_profile_begin(plan.handle, max_rec), "profile_begin")
            t1 = time.perf_counter()
            w = advance(w, args.steps)
            torch.cuda.synchronize(dev)
            elapsed_instr = time.perf_counter() - t1
            tc._lib.check(lib.tcfd_ns2d_profile_end(plan.handle, max_rec, ctypes.byref(cnt), kinds, ms), "profile_end")
        assert torch.isfinite(torch.view_as_real(w)).all().item(), "solution blew up"

    # the same K steps through ONE forward(w, dt, steps=K) call (the reference operator's own `steps` argument):
    # no per-call prologue / dw/dt per step, and with cache-sized chunks the state of a chunk stays on die for all K
    # steps.  Reported next to the headline (which stays per-call, as in round 1), never as `value`.
    fused_api = None
    if not args.fused_steps and world == 1:
        with torch.no_grad():
            op(w, dt, steps=2)
            torch.cuda.synchronize(dev)
            tf = time.perf_counter()
            wf, _ = op(w, dt, steps=args.steps)
            torch.cuda.synchronize(dev)
            tf = time.perf_counter() - tf
            del wf
        fused_api = {"api": "forward(w,dt,steps=K)", "steps_per_s": round(args.steps / tf, 2),
                     "ms_per_step": round(tf / args.steps * 1e3, 3)}

    # ---- the other scaling mode beside the headline: ONE batch of --batch fields cut across the ranks
    strong = None
    if args.scaling == "weak":
        del w
        ws = initial_state(strong_seeds)
        ws, el_s = timed(ws, args.warmup, args.steps)
        el_s_max, el_s_all = over_ranks(el_s)
        del ws
        strong = {"value": round(args.steps / el_s_max, 3), "unit": f"steps/s of ONE batch of {args.batch} fields cut across the GPUs",
                  "batch_total": args.batch, "fields_per_gpu": [b - a for a, b in (shard_batch(args.batch, r, world) for r in range(world))],
                  "ms_per_step": round(el_s_max / args.steps * 1e3, 4),
                  "per_rank_ms_per_step": [round(e / args.steps * 1e3, 4) for e in el_s_all],
                  "note": "efficiency = value(N) / (N * value(1)); no collective inside a step"}
    else:
        del w
    torch.cuda.empty_cache()

    # ---- strong-scaling PROXY on one GPU: what one rank of an N-GPU strong-scaling run does is advance
    # batch/N fields with the same per-call API -- there is no communication inside a step (torch_cfd/equations.py:413-447 is
    # per sample), so t(batch) / t(batch/N) on ONE device IS the N-GPU speed-up up to box-to-box spread and the timing barrier.
    proxy = None
    if world == 1 and args.scaling == "weak" and args.batch >= 8:
        try:
            per_n = {}
            for N in (1, 2, 4, 8):
                if args.batch % N:
                    continue
                wp = initial_state(list(range(args.batch // N)))
                best = None
                for _ in range(3):      # best of three K-step regions: a region at N = 8 is only ~25 ms long
                    wp, el_p = timed(wp, args.warmup, args.steps)
                    best = el_p if best is None else min(best, el_p)
                per_n[N] = best / args.steps * 1e3
                del wp
            torch.cuda.empty_cache()
            proxy = {"fields_per_gpu": {str(N): args.batch // N for N in per_n},
                     "ms_per_step": {str(N): round(t, 4) for N, t in per_n.items()},
                     "predicted_speedup": {str(N): round(per_n[1] / t, 3) for N, t in per_n.items()},
                     "method": f"K x forward(w, dt) on batch/N fields of this ONE GPU, same timed region as `value` (best of 3); "
                               f"speed-up = t({args.batch}) / t({args.batch}/N); exact for this path: no collective inside a step"}
        except Exception as e:
            proxy = {"error": repr(e)}

    # STREAM-style probe of this box: what a plain 16-B/lane copy / read / fill reaches next to the 8 TB/s spec
    probe = {}
    try:
        if args.no_probe:
            raise RuntimeError("skipped (--no-probe)")
        nbytes = 1 << 30
        a_buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        b_buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for mode, name, traffic_x in ((0, "copy_GBps", 2), (1, "read_GBps", 1), (2, "fill_GBps", 1)):
            t_ms = ctypes.c_float(0)
            tc._lib.check(lib.tcfd_hbm_probe(a_buf.data_ptr(), b_buf.data_ptr(), nbytes, mode, 10, ctypes.byref(t_ms), st),
                          "hbm_probe")
            probe[name] = round(traffic_x * nbytes / (t_ms.value * 1e-3) / 1e9, 1)
        del a_buf, b_buf
    except Exception as e:
        probe = {"error": repr(e)}

    per_kind = {}
    for i in range(min(cnt.value, max_rec)):
        per_kind.setdefault(kinds[i], []).append(ms[i])
    # A batched call runs in cache-sized chunks: one full-batch PASS of a kernel is `nchunks` launches.
    # Durations are summed per pass (conservative: back-to-back small launches overlap a little, the sum counts the
    # overlap twice); algorithmic bytes are those of the whole batch, as before.
    rows_per_step = 5
    nchunks = max(1, round(len(per_kind.get(1, [0])) / (args.steps * rows_per_step)))
    kern = {}
    for k, v in per_kind.items():
        avg = sum(v) / len(v)
        ent = {"launches": len(v), "avg_ms": round(avg, 4), "total_ms": round(sum(v), 2)}
        if k in KIND_ALGO_S:
            ent["avg_pass_ms"] = round(avg * nchunks, 4)
            ent["algo_GBps"] = round(KIND_ALGO_S[k] * S / (avg * nchunks * 1e-3) / 1e9, 1)
        kern[KIND_NAMES[k]] = ent
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tpath))
    except Exception:
        tj = {}
    # where a chunk's bytes come from: one chunk's working set (7 workspace fields per batch element) against the
    # 256 MB Infinity Cache -- resident means the per-kernel rates below are L2 <-> Infinity-Cache rates, not DRAM rates
    cf, cb, csrc = ctypes.c_long(0), ctypes.c_size_t(0), ctypes.c_int(0)
    tc._lib.check(lib.tcfd_ns2d_plan_chunking(plan.handle, B, ctypes.byref(cf), ctypes.byref(cb), ctypes.byref(csrc)), "plan_chunking")
    chunking = {"fields_per_chunk": cf.value, "chunks": (-(-B // cf.value) if cf.value else 0), "cache_bytes": cb.value,
                "cache_source": {0: "built-in default", 1: "KFD topology of the device", 2: "TCFD_CACHE_MB"}[csrc.value]}
    resident = 0 < cf.value < B

    def roofline_of(k):
        avg_ms = sum(per_kind[k]) / len(per_kind[k]) * nchunks
        ach = KIND_ALGO_S[k] * S / (avg_ms * 1e-3) / 1e9
        traffic = tj.get(f"{KIND_NAMES[k]}|n{n}|B{B}|{args.dtype}")
        return {"kernel": KIND_NAMES[k],
                "bound": "hbm (algorithmic bytes; the chunk is resident in the 256 MB Infinity Cache)" if resident else "hbm",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                # bytes per launch between L2 and the memory side (Infinity-Cache hits INCLUDED) from rocprofv3 PMC
                # passes of this same command (tests/prof_traffic.py); the file ships with the repo and is NOT
                # re-measured by this run
                "traffic_source": ("profiles/traffic.json: " + str(tj.get("_build", "round-2 build"))) if traffic else None,
                "traffic_level": "L2 <-> fabric (Infinity Cache + HBM); no DRAM-only counter exists on this part" if traffic else None,
                "algo_bytes_per_launch": KIND_ALGO_S[k] * S, "avg_launch_ms": round(avg_ms, 4),
                "launches_per_pass": nchunks,   # avg_launch_ms = one pass over the whole batch = this many chunk launches
                "share_of_step": round(sum(per_kind[k]) / sum(sum(v) for v in per_kind.values()), 3)}

    timed_kinds = [k for k in per_kind if k in KIND_ALGO_S]
    roof = roof_worst = None
    if timed_kinds:
        dom = max(timed_kinds, key=lambda k: sum(per_kind[k]))
        # the kernel furthest below its roofline among those that matter (>= 5 % of the step)
        total_ms = sum(sum(v) for v in per_kind.values())
        worst = min((k for k in timed_kinds if sum(per_kind[k]) >= 0.05 * total_ms),
                    key=lambda k: KIND_ALGO_S[k] / (sum(per_kind[k]) / len(per_kind[k])))
        roof, roof_worst = roofline_of(dom), roofline_of(worst)

    # weak: every rank advanced its own batch -> aggregate batch steps/s; strong: the ranks advanced ONE batch together
    steps_per_s = (world if args.scaling == "weak" else 1) * args.steps / elapsed
    S_job = (world * S) if args.scaling == "weak" else args.batch * n * m * csize   # bytes of one field set, whole job
    # DRAM-level estimate for the chunked step (no counter separates Infinity-Cache hits from DRAM): per forward(w, dt)
    # call every chunk reads its slice of w once and writes w_new and dw/dt once; everything else stays on die
    dram_est = (3.0 if resident else 73.0) * S_job
    out = {
        "metric": baseline_metric(),
        "value": round(steps_per_s, 3),
        "unit": (f"steps/s (one step = all {args.batch} fields of a GPU's batch advance one RK4-CN step)" if args.scaling == "weak"
                 else f"steps/s (one step = the {args.batch} fields of the ONE batch, cut across the GPUs, advance one RK4-CN step)"),
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64" if real == torch.float64 else "f32",
        "data": "synthetic (McWilliams random vorticity, one seed per field, generated on device)",
        "config": {"workload": f"Kolmogorov-forced 2D turbulence, {n}^2 grid, batch {args.batch}{'/GPU' if args.scaling == 'weak' else ' in total'}, "
                               f"RK4-CN pseudo-spectral, nu=1e-3 drag=0.1 sin(4y) forcing dt={dt:.4e}",
                   "n": n, "batch_per_gpu": B if args.scaling == "weak" else None, "batch_total": world * B if args.scaling == "weak" else args.batch,
                   "api": "forward(w,dt,steps=K)" if args.fused_steps else "K x forward(w,dt)",
                   "parallelism": f"batch-sharded x{world}, no data-path collective"},
        "launch": {"spawned_by_bench": os.environ.get("BENCH_SPAWNED") == "1", "process_group": bool(use_dist),
                   "rccl_world_size": dist.get_world_size() if use_dist else None,
                   "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if use_dist else None},
        "per_rank_steps_per_s": [round(args.steps / e, 3) for e in per_rank],
        "sample_steps_per_s": round(args.steps / elapsed * (world * B if args.scaling == "weak" else args.batch), 1),
        "step_algo_GBps": round(70.0 * S_job * args.steps / elapsed / 1e9, 1),
        "step_algo_frac_of_peak": round(70.0 * S_job * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * world), 4),
        "dram_bytes_per_step_est": dram_est,
        "dram_GBps_est": round(dram_est * args.steps / elapsed / 1e9, 1),
        "dram_note": ("physical DRAM traffic of a chunked forward(w, dt) call: read w, write w_new and dw/dt (3 S); the other 70 S "
                      "of the pass model are served by the Infinity Cache" if resident else
                      "not chunked: the pass model's bytes are DRAM bytes"),
        "chunking": chunking,
        "roofline": roof,
        "roofline_worst": roof_worst,
        "kernels": kern,
        "instrumented_ms_per_step": round(elapsed_instr / args.steps * 1e3, 3),
        "fused_steps_api": fused_api,
        "strong_scaling": strong,
        "strong_scaling_proxy": proxy,
        "hbm_probe": probe,
    }
    if not args.no_c4:
        if world > 1 or os.environ.get("BENCH_C4_SEPARATE") == "1":   # (the env switch exercises this path on one GPU)
            # a job of its own (see c4_as_separate_job): rank 0 starts it and waits, the other ranks wait on the host
            torch.cuda.empty_cache()
            if rank == 0:
                out["c4_ensemble"] = c4_as_separate_job(world, args.c4_samples, args.c4_timeout)
            if host_group is not None:
                dist.barrier(group=host_group)
        else:
            try:
                out["c4_ensemble"] = c4_ensemble(dev, world, rank, args.c4_samples)
                if world == 1 and args.c4_samples % 8 == 0 and args.c4_samples >= 64:
                    # one-GPU proxy of the 8-GPU job: what rank 0 of 8 runs (full-size page-locked result, its 1/8 of the
                    # samples); the other ranks do strictly less (no result, no D2H)
                    torch.cuda.empty_cache()
                    if hasattr(torch._C, "_host_emptyCache"):
                        torch._C._host_emptyCache()      # rank 0 of a fresh job page-locks its result anew: no cached blocks
                    one = out["c4_ensemble"]
                    px = c4_ensemble(dev, 1, 0, args.c4_samples, as_rank0_of=8)
                    px["predicted_speedup_whole_job"] = round(one["seconds"] / px["seconds"], 2)
                    px["predicted_speedup_stepping"] = round(one["stepping_s"] / px["stepping_s"], 2)
                    px["method"] = ("this GPU runs what rank 0 of an 8-rank job runs (full-size page-locked result allocated on a "
                                    "helper thread, its 64 of the 512 samples); left out: receiving the 7 peers' records over "
                                    "xGMI and their D2H (7 x 67 MB per record interval on side streams)")
                    out["c4_ensemble"]["strong_scaling_proxy_8gpu"] = px
            except Exception as e:
                out["c4_ensemble"] = {"error": repr(e)}
        torch.set_default_dtype(real)
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_sfno:
        try:
            out["other_configs"] = other_baseline_configs(dev, with_cpu=not args.no_cpu_baseline)
        except Exception as e:
            out["other_configs"] = {"error": repr(e)}
        try:
            out["sfno_config5"] = sfno_config5(dev, with_cpu=not args.no_cpu_baseline)
        except Exception as e:  # secondary measurement: never takes the headline line down
            out["sfno_config5"] = {"error": repr(e)}
        for key, width, act in (("sfno_w16_gelu", 16, "GELU"), ("sfno_w20", 20, "ReLU"), ("sfno_w20_gelu", 20, "GELU"),
                                ("sfno_w32", 32, "ReLU")):
            try:      # the reference's other widths / activation (fno/sfno_pytest.py:261, its notebooks, fno/train.py:303); SURVEY 8d: width 32
                out[key] = sfno_width_line(dev, width, act=act)
            except Exception as e:
                out[key] = {"error": repr(e)}
            torch.cuda.empty_cache()
        try:
            out["sfno_notebook_training"] = sfno_notebook_training(dev)
        except Exception as e:
            out["sfno_notebook_training"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(n, real, dt, args.cpu_seconds)
        except Exception as e:  # the GPU number stands even if the host leg fails
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        # the numbers a reader wants first, LAST in the line (a truncated tail of stdout still shows them) and on stderr
        def pick(d, *path):
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        out["summary"] = {
            "steps_per_s": out["value"], "n_gpus": world, "roofline_frac": pick(out, "roofline", "frac"),
            "step_algo_frac_of_peak": out["step_algo_frac_of_peak"],
            "strong_scaling_proxy_predicted_speedup_8": pick(out, "strong_scaling_proxy", "predicted_speedup", "8"),
            "c4_8gpu_proxy_speedup_whole_job": pick(out, "c4_ensemble", "strong_scaling_proxy_8gpu", "predicted_speedup_whole_job"),
            "c4_sample_steps_per_s": pick(out, "c4_ensemble", "sample_steps_per_s"),
            "C2_steps_per_s": pick(out, "other_configs", "C2_256x16_f32", "steps_per_s"),
            "C4_shard_f64_steps_per_s": pick(out, "other_configs", "C4_shard_512x64_f64", "steps_per_s"),
            "C4_shard_f32_steps_per_s": pick(out, "other_configs", "C4_shard_512x64_f32", "steps_per_s"),
            "C5_forward_plus_loss_ms": pick(out, "sfno_config5", "forward_plus_loss_ms"),
            "C5_frac_of_hbm_peak": pick(out, "sfno_config5", "frac_of_hbm_peak"),
            "C5_train_step_ms": pick(out, "sfno_config5", "train_step_ms"),
            "C5_train_step_ms_gelu": pick(out, "sfno_config5", "train_step_ms_gelu"),
            "w20_train_step_ms": pick(out, "sfno_w20", "train_step_ms"), "w32_train_step_ms": pick(out, "sfno_w32", "train_step_ms"),
            "w32_forward_plus_loss_ms": pick(out, "sfno_w32", "forward_plus_loss_ms"),
            "cpu_baseline_steps_per_s": pick(out, "cpu_baseline", "value"),
            "multi_gpu_curve_measured": world > 1,
        }
        sys.stderr.write("bench summary: " + json.dumps(out["summary"]) + "\n")
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
