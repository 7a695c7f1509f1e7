#!/usr/bin/env python
"""bench_secondary.py -- the secondary workloads beside bench.py's headline, as a job of their own.

bench.py starts this script as a separate process with a time limit AFTER its own measurement is finished (its device
memory released), so nothing here -- an error, a slow host leg, a hang at width 32 -- can delay or take down the headline
record.  What runs: the other BASELINE solver configs (C1, C2, the C4 per-GPU shard in fp64 / fp32, 768^2, 2048^2),
BASELINE configs[4] (SFNO forward + loss, training step, ReLU and GELU), the reference's other model widths and the
notebook-size training loop.  The FULL result (per-kernel tables, notes) is written as JSON to --out; bench.py folds a
few scalars per workload into its one line.  The `cpu_baseline*` functions are the only places that import `oracle`.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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



def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True, help="file the full JSON result is written to")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated subset: solver,c5,widths,notebook")
    args = ap.parse_args(argv)
    assert torch.cuda.is_available(), "bench_secondary.py needs a HIP device"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    only = set(filter(None, args.only.split(",")))
    out = {}

    def flush():
        tmp = args.out + ".tmp"
        with open(tmp, "w") as f:
            json.dump(out, f)
        os.replace(tmp, args.out)      # whatever is finished survives a time limit on the rest

    def leg(key, fn):
        t0 = time.perf_counter()
        try:
            out[key] = fn()
        except Exception as e:         # a secondary measurement never takes the others down
            out[key] = {"error": repr(e)}
        if isinstance(out[key], dict):
            out[key]["wall_s"] = round(time.perf_counter() - t0, 1)
        torch.cuda.empty_cache()
        flush()

    with_cpu = not args.no_cpu_baseline
    if not only or "solver" in only:
        leg("other_configs", lambda: other_baseline_configs(dev, with_cpu=with_cpu))
    if not only or "c5" in only:
        leg("sfno_config5", lambda: sfno_config5(dev, with_cpu=with_cpu))
    if not only or "widths" in only:
        for key, width, act in (("sfno_w16_gelu", 16, "GELU"), ("sfno_w20", 20, "ReLU"), ("sfno_w20_gelu", 20, "GELU"),
                                ("sfno_w32", 32, "ReLU")):
            # the reference's other widths / activation (fno/sfno_pytest.py:261, its notebooks, fno/train.py:303); SURVEY 8d: width 32
            leg(key, lambda: sfno_width_line(dev, width, act=act))
    if not only or "notebook" in only:
        leg("sfno_notebook_training", lambda: sfno_notebook_training(dev))
    flush()


if __name__ == "__main__":
    main()
