"""CPU, world_size 2, gloo: the N > 1 host path (batch sharding + snapshot gather).  The step itself needs a
GPU; here every rank fabricates its shard's records deterministically so the gather can be checked exactly."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_records(start, stop, T=3, n=8):
    m = n // 2 + 1
    idx = torch.arange(start, stop, dtype=torch.float32)[:, None, None, None]
    base = torch.arange(T * n * m, dtype=torch.float32).reshape(1, T, n, m)
    return {k: torch.complex(idx * 10 + base * s, idx - base * s) for k, s in
            (("vorticity", 1.0), ("stream", 0.5), ("vort_t", 2.0), ("residual", -1.0))}


def _worker(rank, world, port, total, q, dst=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_cfd_amd.distributed import gather_trajectory, shard_batch

        a, b = shard_batch(total, rank, world)
        local = _fake_records(a, b) if b > a else {}   # an empty shard holds no tensors at all
        full = gather_trajectory(local, total, dst=dst)
        if rank == dst:
            ref = _fake_records(0, total)
            ok = all(torch.equal(full[k], ref[k]) for k in ref) and sorted(full) == sorted(ref)
            q.put(("ok" if ok else "mismatch", {k: tuple(v.shape) for k, v in full.items()}))
        else:
            q.put(("none" if full is None else "unexpected", None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,dst", [(6, 0), (7, 0), (7, 1), (1, 0), (1, 1)])  # even, ragged, empty shard (either end)
def test_gather_trajectory_world2(total, dst):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q, dst)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    status = sorted(r[0] for r in res)
    assert status == ["none", "ok"], res
    shapes = [r[1] for r in res if r[1]][0]
    assert shapes["vorticity"] == (total, 3, 8, 5)


def test_shard_batch_partitions_exactly():
    from torch_cfd_amd.distributed import shard_batch

    for total in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_batch(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_batch(4, 4, 4)


def test_gather_is_identity_without_process_group():
    from torch_cfd_amd.distributed import gather_trajectory

    rec = _fake_records(0, 2)
    assert gather_trajectory(rec, 2) is rec
