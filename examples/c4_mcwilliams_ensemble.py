#!/usr/bin/env python
"""BASELINE configs[3] ("C4") end to end: a McWilliams decaying-turbulence ensemble, batch-sharded over the GPUs of a node.

    512^2, 64 samples per GPU, fp64 compute, nu = 1e-3, dt = 1e-3, 100 warm-up steps + 550 recorded steps with
    record_every = 55 (10 snapshots x 4 fields), irfft2 + subsample + cast on the device, ONE gather to rank 0.

  one GPU :  python examples/c4_mcwilliams_ensemble.py
  N GPUs  :  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                 examples/c4_mcwilliams_ensemble.py            (weak scaling: 64 samples per GPU)

The drop-in replacement of the batch loop of fno/data_gen/data_gen_McWilliams2d.py:119-171 (`torch_cfd_amd.data_gen`);
no collective inside a step, RCCL only for the final gather (`torch_cfd_amd.distributed.gather_trajectory`).
Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--per-gpu", type=int, default=64)
    ap.add_argument("--warmup-steps", type=int, default=100)
    ap.add_argument("--steps", type=int, default=550)
    ap.add_argument("--record-every", type=int, default=55)
    ap.add_argument("--subsample", type=int, default=2)
    ap.add_argument("--out", default=None, help="torch.save the dataset dict here (rank 0)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torch.distributed.run, also with one rank
    if launched:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    from torch_cfd_amd.data_gen import generate_mcwilliams_dataset

    torch.set_default_dtype(torch.float64)          # data_gen_McWilliams2d.py:103
    total = world * args.per_gpu
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    data = generate_mcwilliams_dataset(args.n, total, args.per_gpu, 1e-3, args.warmup_steps, args.steps, args.record_every,
                                       viscosity=1e-3, peak_wavenumber=4, random_state=0, subsample=args.subsample,
                                       dtype=torch.float32, cdtype=torch.complex64, device=dev, path=args.out)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    if launched:
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = tmax.item()
    if rank == 0:
        nsteps = args.warmup_steps + args.steps
        shapes = {k: list(v.shape) for k, v in data.items()}
        finite = all(bool(torch.isfinite(v).all()) for k, v in data.items() if v.is_floating_point())
        print(json.dumps({"config": f"C4: {args.n}^2, {args.per_gpu} samples/GPU x {world} GPU, fp64, {args.warmup_steps}+{args.steps} steps, "
                                    f"record every {args.record_every}", "n_gpus": world, "process_group": bool(launched), "seconds": round(el, 3),
                          "sample_steps_per_s": round(total * nsteps / el, 1), "batch_steps_per_s_per_gpu": round(nsteps / el, 1),
                          "shapes": shapes, "finite": finite}), flush=True)
    if launched:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
