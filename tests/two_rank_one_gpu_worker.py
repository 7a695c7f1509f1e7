"""Worker of tests/test_launch_gpu.py::test_two_ranks_share_the_one_device: one of two ranks that BOTH use cuda:0.
mode "rccl":   init the RCCL group and try a point-to-point exchange (expected to be refused: duplicate GPU).
mode "staged": the record hand-over over a gloo group with DEVICE tensors (RecordHandover.staged) -- fabricated records, then
               the real ensemble job (generate_mcwilliams_dataset) whose result rank 0 writes to <out>."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fake_packed(start, count, rec, dev, F=4, ns=8):
    idx = torch.arange(start, start + count, dtype=torch.float32)[:, None, None, None]
    f = torch.arange(F, dtype=torch.float32)[None, :, None, None]
    yx = torch.arange(ns * ns, dtype=torch.float32).reshape(1, 1, ns, ns)
    return (idx * 1000 + rec * 100 + f * 10 + yx / 64).contiguous().to(dev)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if mode == "rccl":
        import datetime

        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=60))
        x = torch.full((1024,), float(rank), device=dev)
        ops = [dist.P2POp(dist.isend if rank else dist.irecv, x, 1 - rank)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize()
        print("RCCL_TWO_RANKS_ONE_DEVICE_OK", float(x[0]))
        dist.destroy_process_group()
        return
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_cfd_amd.data_gen import generate_mcwilliams_dataset
    from torch_cfd_amd.distributed import TRAJECTORY_FIELDS, RecordHandover, batch_layout

    # 1. fabricated records: several batches per rank, ragged, dst = 1 (the peer is rank 0)
    total, batch, n_rec, dst = 7, 2, 3, 1
    layout = batch_layout(total, world, batch)
    ho = RecordHandover(TRAJECTORY_FIELDS, total, n_rec, (8, 8), torch.float32, layout, dev, dst=dst)
    assert ho.staged and ho.on_gpu
    for start, count in layout[rank]:
        for rec in range(n_rec):
            ho.push(start, rec, fake_packed(start, count, rec, dev))
    full = ho.finish()
    if rank == dst:
        for f, name in enumerate(TRAJECTORY_FIELDS):
            for rec in range(n_rec):
                assert torch.equal(full[name][:, rec], fake_packed(0, total, rec, "cpu")[:, f]), (name, rec)
    else:
        assert full is None
    # 2. the real job: 6 samples of 64^2 in batches of 2, cut across the two ranks; rank 0 saves the dataset
    torch.set_default_dtype(torch.float64)
    data = generate_mcwilliams_dataset(64, 6, 2, 1e-3, 4, 12, 4, random_state=3, subsample=2, device=dev)
    if rank == 0:
        torch.save(data, out)
    else:
        assert data is None
    dist.barrier()
    dist.destroy_process_group()
    print("STAGED_OK")


if __name__ == "__main__":
    main()
