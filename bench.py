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

Prints ONE JSON line (rank 0), under 8 KB, and nothing holding a brace after it on
either stream (stderr is passed through a filter that replaces braces).  `value`
is the median of `--regions` regions of K timed steps.  `roofline` is for the
dominant kernel, measured with HIP events recorded by the library on the launch
stream over K instrumented steps; `cpu_baseline` times the CPU oracle (a torch-CPU
restatement of the reference's op sequence) on this host.  The secondary
workloads (other solver configs, the SFNO model runs) are a separate process with
a time limit (bench_secondary.py) started after the headline is measured; a few
scalars of each are folded into the line, the full record goes to
gpurun_out/bench_detail.json.
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
    ap.add_argument("--regions", type=int, default=5, help="timed regions of W + K steps; value = the median region")
    ap.add_argument("--preheat", type=int, default=20, help="untimed steps before the first region (clock ramp, caches)")
    ap.add_argument("--secondary-timeout", type=float, default=420.0,
                    help="seconds the separate job of secondary workloads (bench_secondary.py) may take")
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
           "roofline": None, "cpu_baseline": None,
           "dry_run": True, "backend": "gloo", "world_size": dist.get_world_size() if dist.is_initialized() else 1,
           "fields_of_rank0": [lo, hi], "fields_of_every_rank": spans, "per_rank_steps_per_s": [round(r, 1) for r in rates],
           "spawned_by_bench": os.environ.get("BENCH_SPAWNED") == "1"}
    if os.environ.get("BENCH_TEST_STDERR_NOISE") == "1":     # tests/test_host_logic.py: braces on both streams, before and after the line
        print('{"noise": "python stdout"}')
        print('warning: {"noise": "python stderr"}', file=sys.stderr)
        os.write(2, b'{"noise": "fd 2"}\n')
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if os.environ.get("BENCH_TEST_STDERR_NOISE") == "1":
        os.write(2, b'bench summary: {"noise": "after the line"}\n')
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
            "handover_mode": stats.get("handover_mode"),
            "note": "max over ranks of each phase; handover_tail_s = gather + D2H left after the last step finished"}


LAUNCH_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
              "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "BENCH_SPAWNED", "BENCH_FORCE_DIST", "TORCHELASTIC_RUN_ID",
              "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")


def c4_as_separate_job(world, samples, timeout):
    """The C4 ensemble job of a multi-rank run, started by rank 0 as its OWN set of `world` ranks (this script with --c4-only,
    through the spawner above) with a time limit: its hand-over uses point-to-point RCCL traffic that the step benchmark does
    not, and whatever happens to it -- an error, a hang -- must not take the scaling measurement down with it.  The parent
    ranks idle meanwhile (their GPUs are shared with the job: a few GB of their 288).
    The hand-over's point-to-point traffic is among a SUBSET of the communicator's ranks (a peer and rank 0), the one RCCL
    call pattern no test of this repository has seen with a real peer: if the job fails or runs out of time in that mode
    (and the caller did not pick a mode), it is run ONCE more with one gather per record interval on every rank
    (TCFD_HANDOVER=collective, torch-cfd_amd/distributed.py); `handover_mode` / `first_attempt` say what happened."""
    import signal
    import subprocess

    def attempt(extra_env):
        env = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV}
        env.update(extra_env)
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

    res = attempt({})
    if "error" in res and world > 1 and not os.environ.get("TCFD_HANDOVER"):
        again = attempt({"TCFD_HANDOVER": "collective"})
        again["first_attempt"] = {"handover_mode": "p2p", "error": res["error"]}
        again.setdefault("handover_mode", "gather")
        return again
    return res


def sanitize_stderr():
    """Everything this process (and the libraries under it) writes to fd 1 / fd 2 from here on reaches the real stderr
    with braces replaced by parentheses.  The record of a run is the LAST line holding a JSON object -- the one line on
    stdout; a later brace on stderr (a dict in a warning, a summary) used to make a parser pick the wrong line
    (BENCH_r05: parsed = null).  A pipe and a pump thread, so the text still appears while the run goes on.
    Returns (fd of the real stdout, function that drains and restores)."""
    import fcntl
    import threading

    sys.stdout.flush(); sys.stderr.flush()
    real_stdout, real_stderr = os.dup(1), os.dup(2)
    rd, wr = os.pipe()
    try:
        fcntl.fcntl(wr, 1031, 1 << 20)      # F_SETPIPE_SZ: room for a burst written while the GIL is held elsewhere
    except OSError:
        pass
    os.dup2(wr, 1); os.dup2(wr, 2); os.close(wr)
    table = bytes.maketrans(b"{}", b"()")

    def pump():
        while True:
            try:
                chunk = os.read(rd, 65536)
            except OSError:
                break
            if not chunk:
                break
            os.write(real_stderr, chunk.translate(table))

    th = threading.Thread(target=pump, daemon=True)
    th.start()

    def restore():
        sys.stdout.flush(); sys.stderr.flush()
        os.dup2(real_stderr, 2); os.dup2(real_stderr, 1)      # closes the last write ends of the pipe: the pump sees EOF
        th.join(timeout=5)
    import atexit
    atexit.register(restore)
    return real_stdout, restore


def secondary_job(timeout, with_cpu=True, only=""):
    """bench_secondary.py as its own process with a time limit; returns its JSON (what it finished) or {"error": ...}."""
    import signal
    import subprocess
    import tempfile

    env = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV}
    fd, path = tempfile.mkstemp(suffix=".json", prefix="bench_secondary_")
    os.close(fd)
    cmd = [sys.executable, os.path.join(ROOT, "bench_secondary.py"), "--out", path]
    if not with_cpu:
        cmd.append("--no-cpu-baseline")
    if only:
        cmd += ["--only", only]
    res, note = {}, None
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=sys.stderr, stderr=sys.stderr, start_new_session=True)
        try:
            proc.wait(timeout=timeout)
            if proc.returncode != 0:
                note = f"bench_secondary.py exited with status {proc.returncode}"
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)      # the job's own process group (start_new_session)
            proc.wait()
            note = f"bench_secondary.py did not finish within {timeout:.0f} s and was stopped; finished legs are kept"
        try:
            res = json.load(open(path))
        except Exception:
            res = {}
    except Exception as e:
        note = repr(e)
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    if note:
        res["error"] = note
    return res


def _pick(d, *path):
    for k in path:
        d = d.get(k) if isinstance(d, dict) else None
    return d


def _few(d, keys):
    """{key: d[key]} for the keys that are present and scalar (or an error string)."""
    if not isinstance(d, dict):
        return None
    out = {k: d[k] for k in keys if isinstance(d.get(k), (int, float, str, bool))}
    if "error" in d:
        out["error"] = str(d["error"])[:160]
    return out


def compose_line(out, detail):
    """(the ONE stdout line, the full detail record).  The line carries the contract's keys, `roofline`, `cpu_baseline`, the
    per-kernel table of the headline and at most six scalars per secondary workload, `summary` last; it must stay under
    8 KB so that any tail of the output holds all of it (tests/test_host_logic.py).  Everything else -- per-kernel tables
    of the model runs, per-region / per-rank times, the notes -- goes to the detail file."""
    line = dict(out)
    full = dict(out)
    full.update(detail)
    sec = detail.get("secondary") or {}
    for k in ("roofline", "roofline_worst"):
        if isinstance(out.get(k), dict):
            line[k] = {kk: out[k][kk] for kk in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algo_bytes_per_launch",
                                                 "avg_launch_ms", "launches_per_pass", "share_of_step") if kk in out[k]}
            line[k]["bound"] = str(line[k].get("bound", "hbm")).split(" ")[0]      # "hbm" | "mfma"; the remark is in the detail record
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "threads", "cpu_model", "kind", "sample", "error") if k in cb}
    if isinstance(out.get("strong_scaling"), dict):
        line["strong_scaling"] = dict(_few(out["strong_scaling"], ("value", "ms_per_step", "batch_total")),
                                      fields_per_gpu=out["strong_scaling"].get("fields_per_gpu"))
    if isinstance(out.get("strong_scaling_proxy"), dict):
        line["strong_scaling_proxy"] = {k: v for k, v in out["strong_scaling_proxy"].items() if k in ("predicted_speedup", "ms_per_step", "error")}
    c4 = out.get("c4_ensemble")
    if isinstance(c4, dict):
        line["c4_ensemble"] = _few(c4, ("seconds", "sample_steps_per_s", "stepping_s", "handover_tail_s", "dataset_GB", "finite", "handover_mode"))
        if isinstance(c4.get("first_attempt"), dict):
            line["c4_ensemble"]["first_attempt"] = _few(c4["first_attempt"], ("handover_mode",))
        px = c4.get("strong_scaling_proxy_8gpu")
        if isinstance(px, dict):
            line["c4_ensemble"]["proxy_8gpu"] = _few(px, ("seconds", "stepping_s", "predicted_speedup_whole_job", "predicted_speedup_stepping"))
    if sec:
        oc = sec.get("other_configs") or {}
        line["other_configs"] = {name: dict(_few(v, ("steps_per_s", "ms_per_step", "step_algo_GBps", "gpu_over_cpu")) or {},
                                            **({"cpu_steps_per_s": round(v["cpu_baseline"]["value"], 3)}
                                               if _pick(v, "cpu_baseline", "value") else {}))
                                 for name, v in oc.items() if isinstance(v, dict)} or _few(oc, ())
        c5 = sec.get("sfno_config5")
        if isinstance(c5, dict):
            line["sfno_config5"] = _few(c5, ("forward_ms", "forward_plus_loss_ms", "frac_of_hbm_peak", "train_step_ms", "train_step_ms_gelu",
                                              "forward_ms_with_gelu", "gpu_over_cpu"))
            r5 = c5.get("roofline")
            if isinstance(r5, dict):
                line["sfno_config5"]["roofline"] = _few(r5, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"))
                bk = r5.get("backward_kernel")
                if isinstance(bk, dict):
                    line["sfno_config5"]["backward_kernel"] = dict(_few(bk, ("avg_launch_ms", "frac")) or {},
                                                                   mfma_frac=_pick(bk, "mfma", "frac"), mfma_useful_frac=_pick(bk, "mfma", "useful_frac"))
            if _pick(c5, "cpu_baseline", "value"):
                line["sfno_config5"]["cpu_samples_per_s"] = round(c5["cpu_baseline"]["value"], 3)
        for key in ("sfno_w16_gelu", "sfno_w20", "sfno_w20_gelu", "sfno_w32"):
            if isinstance(sec.get(key), dict):
                line[key] = _few(sec[key], ("forward_ms", "forward_plus_loss_ms", "train_step_ms", "frac_of_hbm_peak", "train_step_ms_per_GB",
                                            "einsum_recompute_fallbacks_in_a_training_step"))
        nb = sec.get("sfno_notebook_training")
        if isinstance(nb, dict):
            line["sfno_notebook_training"] = dict(_few(nb, ("iterations_per_s", "ms_per_iteration", "finite")) or {},
                                                  graph_replay_iterations_per_s=_pick(nb, "graph_replay", "iterations_per_s"))
        if "error" in sec:
            line["secondary_error"] = str(sec["error"])[:200]
    # the numbers a reader wants first, LAST in the line
    line["summary"] = {
        "steps_per_s": out.get("value"), "n_gpus": out.get("n_gpus"), "roofline_frac": _pick(out, "roofline", "frac"),
        "roofline_worst_frac": _pick(out, "roofline_worst", "frac"),
        "step_algo_frac_of_peak": out.get("step_algo_frac_of_peak"),
        "strong_scaling_proxy_predicted_speedup_8": _pick(out, "strong_scaling_proxy", "predicted_speedup", "8"),
        "c4_8gpu_proxy_speedup_whole_job": _pick(out, "c4_ensemble", "strong_scaling_proxy_8gpu", "predicted_speedup_whole_job"),
        "C2_steps_per_s": _pick(sec, "other_configs", "C2_256x16_f32", "steps_per_s"),
        "C4_shard_f64_steps_per_s": _pick(sec, "other_configs", "C4_shard_512x64_f64", "steps_per_s"),
        "C4_shard_f32_steps_per_s": _pick(sec, "other_configs", "C4_shard_512x64_f32", "steps_per_s"),
        "C5_forward_plus_loss_ms": _pick(sec, "sfno_config5", "forward_plus_loss_ms"),
        "C5_train_step_ms": _pick(sec, "sfno_config5", "train_step_ms"),
        "cpu_baseline_steps_per_s": _pick(out, "cpu_baseline", "value"),
        "multi_gpu_curve_measured": (out.get("n_gpus") or 1) > 1,
    }
    full["summary"] = line["summary"]
    return line, full


def write_detail(full):
    """The full record beside the line: gpurun_out/bench_detail.json (that directory comes back from a GPU box), else the
    repo root.  Returns the path or None."""
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_detail.json")
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            return path
        except OSError:
            continue
    return None


def c4_only_main(args):
    """`--c4-only`: the C4 job alone, one JSON object on stdout (rank 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    real_stdout, _restore = sanitize_stderr()
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
    real_stdout, _restore = sanitize_stderr()
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
        if k <= 0:
            return w
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

    # The measurement: after `--preheat` untimed steps (clocks and caches in their steady state), `--regions` regions of
    # W untimed + EXACTLY K timed steps each, every region between two fences (barrier + synchronize) and reduced with
    # max over ranks; `value` / `ms_per_step` are those of the MEDIAN region, all regions are listed in `regions_ms_per_step`.
    # One region of 20 steps is 0.15 s: a single one moved the headline by 10 % from lease to lease (round 5).
    # Then ONE instrumented pass of K steps: the library brackets every launch with HIP events on the launch stream.
    # It is separate because a chunked step is ~176 short launches and an event pair per launch costs ~7 us of host
    # time and in-order queue barriers each (measured: 7.56 -> 8.78 ms per step) -- bracketing the timed regions
    # themselves would slow down the very number they produce.  `instrumented_ms_per_step` shows the difference.
    with torch.no_grad():
        w = advance(w, args.preheat)
    regions = []
    for _ in range(max(1, args.regions)):
        w, el_local = timed(w, args.warmup, args.steps)
        regions.append(over_ranks(el_local))
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    elapsed, per_rank = regions[order[(len(order) - 1) // 2]]      # median (the lower one of an even count)
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
        # the same three loops on 2 x 96 MB: buffers that stay in the 256 MB Infinity Cache, as a chunk of the step does (its
        # 239 MB working set) -- the ceiling the per-kernel algorithmic rates of a chunked step are to be read against
        res_bytes = 96 << 20
        for mode, name, traffic_x in ((0, "resident_copy_GBps", 2), (1, "resident_read_GBps", 1), (2, "resident_fill_GBps", 1)):
            t_ms = ctypes.c_float(0)
            tc._lib.check(lib.tcfd_hbm_probe(a_buf.data_ptr(), b_buf.data_ptr(), res_bytes, mode, 20, ctypes.byref(t_ms), st),
                          "hbm_probe")
            probe[name] = round(traffic_x * res_bytes / (t_ms.value * 1e-3) / 1e9, 1)
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
                "bound": "hbm",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algo_bytes_per_launch": KIND_ALGO_S[k] * S, "avg_launch_ms": round(avg_ms, 4),
                "launches_per_pass": nchunks,   # avg_launch_ms = one pass over the whole batch = this many chunk launches
                "share_of_step": round(sum(per_kind[k]) / sum(sum(v) for v in per_kind.values()), 3)}

    roofline_notes = {
        "bound": ("algorithmic bytes against the HBM peak; a chunk's working set is resident in the 256 MB Infinity Cache, so the "
                  "physical DRAM traffic is lower (see dram_*)") if resident else "algorithmic bytes = DRAM bytes (not chunked)",
        # bytes per launch between L2 and the memory side (Infinity-Cache hits INCLUDED) from rocprofv3 PMC passes of this same
        # command (tests/prof_traffic.py); the file ships with the repo and is NOT re-measured by this run
        "traffic_source": "profiles/traffic.json: " + str(tj.get("_build", "round-2 build")),
        "traffic_level": "L2 <-> fabric (Infinity Cache + HBM); no DRAM-only counter exists on this part",
        "timed": "HIP events recorded by the library around every launch on its launch stream, one instrumented pass of K steps"}

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
    per_region = [round(e / args.steps * 1e3, 4) for e, _ in regions]
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
        "timing": {"value_is": f"median of {len(regions)} regions of W untimed + K timed steps, each between barrier + synchronize, max over ranks",
                   "regions_ms_per_step": per_region, "preheat_steps": args.preheat,
                   "spread": round((max(per_region) - min(per_region)) / min(per_region), 4)},
        "launch": {"spawned_by_bench": os.environ.get("BENCH_SPAWNED") == "1", "process_group": bool(use_dist),
                   "rccl_world_size": dist.get_world_size() if use_dist else None,
                   "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if use_dist else None},
        "per_rank_steps_per_s": [round(args.steps / e, 3) for e in per_rank],
        "sample_steps_per_s": round(args.steps / elapsed * (world * B if args.scaling == "weak" else args.batch), 1),
        "step_algo_GBps": round(70.0 * S_job * args.steps / elapsed / 1e9, 1),
        "step_algo_frac_of_peak": round(70.0 * S_job * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * world), 4),
        "dram_bytes_per_step_est": dram_est,
        "dram_GBps_est": round(dram_est * args.steps / elapsed / 1e9, 1),
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
    detail = {"roofline_notes": roofline_notes,
              "dram_note": ("physical DRAM traffic of a chunked forward(w, dt) call: read w, write w_new and dw/dt (3 S); the other 70 S "
                            "of the pass model are served by the Infinity Cache" if resident else
                            "not chunked: the pass model's bytes are DRAM bytes"),
              "regions": [{"seconds": e, "per_rank_seconds": pr} for e, pr in regions]}
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
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(n, real, dt, args.cpu_seconds)
        except Exception as e:  # the GPU number stands even if the host leg fails
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and world == 1 and not args.no_sfno:
        # everything that is not the headline runs as a job of its own with a time limit, after this process has released
        # its device memory: nothing in there can delay or break the record above
        del op, plan, plan_fft
        torch.cuda.empty_cache()
        detail["secondary"] = secondary_job(args.secondary_timeout, with_cpu=not args.no_cpu_baseline,
                                            only=os.environ.get("BENCH_SECONDARY_ONLY", ""))     # (the env switch: a quick subset, for the tests)
    if rank == 0:
        line, detail_all = compose_line(out, detail)
        path = write_detail(detail_all)
        if path:
            line["detail_file"] = os.path.relpath(path, ROOT)
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
