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
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=120)
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


# ----------------------------------------------------------------------------- per-record hand-over (RecordHandover)
def _fake_packed(start, count, rec, F=4, ns=4):
    idx = torch.arange(start, start + count, dtype=torch.float32)[:, None, None, None]
    f = torch.arange(F, dtype=torch.float32)[None, :, None, None]
    yx = torch.arange(ns * ns, dtype=torch.float32).reshape(1, 1, ns, ns)
    return (idx * 1000 + rec * 100 + f * 10 + yx / 16).contiguous()


def _handover_worker(rank, world, port, total, batch, n_rec, q, dst):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_cfd_amd.distributed import TRAJECTORY_FIELDS, RecordHandover, batch_layout

        layout = batch_layout(total, world, batch)
        ho = RecordHandover(TRAJECTORY_FIELDS, total, n_rec, (4, 4), torch.float32, layout, "cpu", dst=dst)
        for start, count in layout[rank]:
            for rec in range(n_rec):
                ho.push(start, rec, _fake_packed(start, count, rec))
        full = ho.finish()
        if rank == dst:
            ok = True
            for f, name in enumerate(TRAJECTORY_FIELDS):
                for rec in range(n_rec):
                    ok &= torch.equal(full[name][:, rec], _fake_packed(0, total, rec)[:, f])
            q.put(("ok" if ok else "mismatch", tuple(full["vorticity"].shape)))
        else:
            q.put(("none" if full is None else "unexpected", None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,batch,dst", [(8, 2, 0), (7, 2, 1), (5, 8, 0), (1, 4, 1)])  # several batches, ragged, empty shard
def test_record_handover_world2(total, batch, dst):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_handover_worker, args=(r, 2, port, total, batch, 3, q, dst)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == ["none", "ok"], res
    assert [r[1] for r in res if r[1]][0] == (total, 3, 4, 4)


@pytest.mark.parametrize("total,batch,dst", [(37, 2, 0), (37, 2, 5), (16, 1, 0), (5, 1, 5)])
def test_record_handover_world8(total, batch, dst):
    """The BASELINE config-4 layout scaled down (fno/data_gen/data_gen_McWilliams2d.py:119-174: 512 samples in batches of 64 over
    8 GPUs): 8 ranks, >= 2 batches per rank, a ragged last batch / last rank (37 = 8 x 4 + 5), ranks with NO sample (5 over 8),
    the result at rank 0 or at an inner rank."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_handover_worker, args=(r, world, port, total, batch, 3, q, dst)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == ["none"] * 7 + ["ok"], res
    assert [r[1] for r in res if r[1]][0] == (total, 3, 4, 4)


def _handover_modes_worker(rank, world, port, cases, n_rec, q):
    """Every case (total, batch, dst) through the three hand-over modes on ONE process group: the subset point-to-point form,
    one batched point-to-point call per interval on every rank, one gather per interval on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_cfd_amd.distributed import TRAJECTORY_FIELDS, RecordHandover, batch_layout

        verdicts = []
        for total, batch, dst in cases:
            layout = batch_layout(total, world, batch)
            got = {}
            for mode in ("p2p", "uniform", "collective"):
                ho = RecordHandover(TRAJECTORY_FIELDS, total, n_rec, (4, 4), torch.float32, layout, "cpu", dst=dst, mode=mode)
                assert ho.mode == {"collective": "gather"}.get(mode, mode)
                for start, count in layout[rank]:
                    for rec in range(n_rec):
                        ho.push(start, rec, _fake_packed(start, count, rec))
                got[mode] = ho.finish()
                dist.barrier()
            if rank == dst:
                want = [torch.equal(got["p2p"][name][:, rec], _fake_packed(0, total, rec)[:, f])
                        for f, name in enumerate(TRAJECTORY_FIELDS) for rec in range(n_rec)]
                same = [torch.equal(got["p2p"][name], got[m][name]) for name in TRAJECTORY_FIELDS for m in ("uniform", "collective")]
                verdicts.append("ok" if all(want) and all(same) else "mismatch")
            else:
                verdicts.append("none" if all(v is None for v in got.values()) else "unexpected")
        q.put(verdicts)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cases", [(2, [(7, 2, 1), (1, 4, 0)]),
                                         (8, [(37, 2, 5), (5, 1, 0), (16, 1, 3)])])   # ragged batches, ranks without a sample
def test_record_handover_group_call_modes_equal_point_to_point(world, cases, monkeypatch):
    """TCFD_HANDOVER = uniform / collective (the fall-backs for a collective library that rejects point-to-point traffic among a
    subset of a communicator's ranks): every rank makes one group call per record interval; the host
    result is bit-equal to the point-to-point mode's, with ragged last batches, ranks that run out of records early and ranks
    that never had one."""
    monkeypatch.delenv("TCFD_HANDOVER", raising=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_handover_modes_worker, args=(r, world, port, cases, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for i in range(len(cases)):
        assert sorted(v[i] for v in res) == ["none"] * (world - 1) + ["ok"], (cases[i], res)


def test_record_handover_mode_from_the_environment(monkeypatch):
    from torch_cfd_amd.distributed import RecordHandover, batch_layout

    monkeypatch.setenv("TCFD_HANDOVER", "collective")
    ho = RecordHandover(("a",), 2, 1, (4, 4), torch.float32, batch_layout(2, 1, 2), "cpu")
    assert ho.mode == "p2p"                 # no process group: nothing to make uniform
    monkeypatch.setenv("TCFD_HANDOVER", "broadcast")
    with pytest.raises(ValueError, match="hand-over mode"):
        RecordHandover(("a",), 2, 1, (4, 4), torch.float32, batch_layout(2, 1, 2), "cpu")



def _handover_subgroup_worker(rank, world, port, total, batch, n_rec, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_cfd_amd.distributed import TRAJECTORY_FIELDS, RecordHandover, batch_layout

        members = [1, 3, 4, 6]                  # a job on four of the eight ranks; the result at GLOBAL rank 4 (group rank 2)
        group = dist.new_group(members)
        if rank not in members:
            q.put(("outside", None))
            return
        grank, gworld = dist.get_rank(group), len(members)
        layout = batch_layout(total, gworld, batch)
        ho = RecordHandover(TRAJECTORY_FIELDS, total, n_rec, (4, 4), torch.float32, layout, "cpu", dst=2, group=group)
        for start, count in layout[grank]:
            for rec in range(n_rec):
                ho.push(start, rec, _fake_packed(start, count, rec))
        full = ho.finish()
        if grank == 2:
            ok = all(torch.equal(full[name][:, rec], _fake_packed(0, total, rec)[:, f])
                     for f, name in enumerate(TRAJECTORY_FIELDS) for rec in range(n_rec))
            q.put(("ok" if ok else "mismatch", tuple(full["vorticity"].shape)))
        else:
            q.put(("none" if full is None else "unexpected", None))
    finally:
        dist.destroy_process_group()


def test_record_handover_in_a_subgroup_of_world8():
    world, total = 8, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_handover_subgroup_worker, args=(r, world, port, total, 2, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == ["none"] * 3 + ["ok"] + ["outside"] * 4, res


def test_record_handover_without_process_group_and_order_check():
    from torch_cfd_amd.distributed import RecordHandover, batch_layout

    layout = batch_layout(5, 1, 2)
    assert layout == [[(0, 2), (2, 2), (4, 1)]]
    ho = RecordHandover(("a", "b"), 5, 2, (4, 4), torch.float32, layout, "cpu")
    with pytest.raises(ValueError, match="out of order"):
        ho.push(2, 0, _fake_packed(2, 2, 0, F=2))
    for start, count in layout[0]:
        for rec in range(2):
            ho.push(start, rec, _fake_packed(start, count, rec, F=2))
    full = ho.finish()
    assert torch.equal(full["b"][:, 1], _fake_packed(0, 5, 1, F=2)[:, 1])
    ho2 = RecordHandover(("a",), 2, 1, (4, 4), torch.float32, batch_layout(2, 1, 2), "cpu")
    with pytest.raises(RuntimeError, match="never pushed"):
        ho2.finish()


def test_record_handover_close_is_idempotent_and_safe_before_finish():
    """A hand-over dropped before ``finish`` (an exception in the stepping loop) releases what it holds: ``close`` after a few
    pushes, twice, and then ``finish`` reports the records that never came instead of hanging."""
    from torch_cfd_amd.distributed import RecordHandover, batch_layout

    layout = batch_layout(5, 1, 2)
    ho = RecordHandover(("a", "b"), 5, 2, (4, 4), torch.float32, layout, "cpu")
    ho.push(0, 0, _fake_packed(0, 2, 0, F=2))
    ho.close()
    ho.close()
    with pytest.raises(RuntimeError, match="never pushed"):
        ho.finish()
    del ho


def _mismatch_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_cfd_amd.distributed import gather_trajectory, shard_batch

        a, b = shard_batch(6, rank, world)
        local = _fake_records(a, b + (1 if rank == 1 else 0))   # rank 1 holds one sample too many
        try:
            gather_trajectory(local, 6)
            q.put("no error")
        except ValueError as e:
            q.put("raised" if "shard size" in str(e) else repr(e))
    finally:
        dist.destroy_process_group()


def test_gather_shard_size_mismatch_raises_on_every_rank():
    """A bad shard must not leave the other ranks blocked in their point-to-point calls."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mismatch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == ["raised", "raised"], res


def _subgroup_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_cfd_amd.distributed import gather_trajectory, shard_batch

        group = dist.new_group([1, 2])     # group ranks 0, 1 = global ranks 1, 2
        if rank == 0:
            q.put("outside")
            return
        grank = dist.get_rank(group)
        a, b = shard_batch(5, grank, 2)
        full = gather_trajectory(_fake_records(a, b), 5, dst=1, group=group)
        if grank == 1:
            ref = _fake_records(0, 5)
            q.put("ok" if all(torch.equal(full[k], ref[k]) for k in ref) else "mismatch")
        else:
            q.put("none" if full is None else "unexpected")
    finally:
        dist.destroy_process_group()


def test_gather_in_a_subgroup_addresses_peers_by_global_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=400) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == ["none", "ok", "outside"], res


# ----------------------------------------------------------------------------- bench.py --gpus N starts its own ranks
def _run_bench(args):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, cwd=root, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    return r, [ln for ln in r.stdout.decode().splitlines() if ln.strip()]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_spawns_its_own_ranks(scaling):
    """`python bench.py --gpus 2` with no launcher: two ranks, gloo, no kernels (--host-only), ONE JSON line."""
    import json

    r, lines = _run_bench(["--gpus", "2", "--steps", "4", "--host-only", "--scaling", scaling])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["spawned_by_bench"] is True and out["dry_run"] is True
    assert out["scaling"] == scaling and len(out["per_rank_steps_per_s"]) == 2
    assert out["fields_of_rank0"] == ([0, 64] if scaling == "weak" else [0, 32])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_spawns_eight_ranks(scaling):
    """What the driver's SCALE run does first: `python bench.py --gpus 8` through the script's own spawner (gloo, --host-only:
    no kernels) -- eight ranks rendezvous, ONE JSON line, every rank took the span `shard_batch` gives it."""
    import json

    from torch_cfd_amd.distributed import shard_batch

    r, lines = _run_bench(["--gpus", "8", "--steps", "3", "--host-only", "--scaling", scaling])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["world_size"] == 8 and out["spawned_by_bench"] is True
    assert len(out["per_rank_steps_per_s"]) == 8
    want = ([list(shard_batch(64, k, 8)) for k in range(8)] if scaling == "strong" else [[64 * k, 64 * (k + 1)] for k in range(8)])
    assert out["fields_of_every_rank"] == want


def test_bench_refuses_more_gpus_than_visible():
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        pytest.skip("this box really has 64 devices")
    r, lines = _run_bench(["--gpus", "64", "--steps", "1"])
    assert r.returncode == 2 and not lines and b"HIP device" in r.stderr
