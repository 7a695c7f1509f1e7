"""Batch-sharded ensembles over the GPUs of one node.

Every batch element of the spectral solver is an independent trajectory (all operators act on the last two
dims), so the path shards with NO collective inside a step: each rank owns a contiguous slice of the batch, its
own plan and workspace.  The only data-path communication is the hand-over of recorded snapshots to one rank at
the end of a trajectory: point-to-point sends straight into the slices of ONE pre-allocated result on ``dst``
(RCCL over xGMI for HIP tensors -- seven peers send over seven distinct links in parallel; gloo on CPU in the
tests).  Nobody but ``dst`` allocates anything, shards may be ragged or empty.  The reference has no distributed
code; this is what its data-generation drivers need
(fno/data_gen/data_gen_McWilliams2d.py:126-152 loops over batches serially on one device).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

TRAJECTORY_FIELDS = ("vorticity", "stream", "vort_t", "residual")


def shard_batch(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[start, stop) of ``rank``'s contiguous slice of ``total`` batch elements (sizes differ by at most 1)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(total, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _global_rank(group, group_rank: int) -> int:
    """``dist.P2POp`` / ``isend`` / ``irecv`` address peers by GLOBAL rank; ranks and ``dst`` of this module are
    ranks within ``group``."""
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def _describe(local: Dict[str, torch.Tensor]) -> List[tuple]:
    return [(k, tuple(v.shape[1:]), str(v.dtype)) for k, v in sorted(local.items())]


def gather_trajectory(local: Dict[str, torch.Tensor], total_batch: int, dst: int = 0,
                      group: Optional[dist.ProcessGroup] = None,
                      keys: Optional[Sequence[str]] = None) -> Optional[Dict[str, torch.Tensor]]:
    """Concatenate per-rank dicts ``{key: (B_rank, ...)}`` along the batch axis on rank ``dst``.

    Every rank takes part with the SAME key list: ``keys`` if given, else the description (key, trailing shape,
    dtype) of the first non-empty shard, agreed on with one small object collective -- a rank whose shard is empty
    (``total_batch < world_size``) holds no tensors and still has to know what the others send.  Returns the
    assembled dict on ``dst`` and ``None`` elsewhere; without a process group the input is returned unchanged."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    spans = [shard_batch(total_batch, r, world) for r in range(world)]
    lo, hi = spans[rank]
    mine = _describe(local) if keys is None else _describe({k: local[k] for k in keys if k in local})
    sizes = sorted({int(v.shape[0]) for v in local.values()})
    every: List[Optional[tuple]] = [None] * world
    dist.all_gather_object(every, (mine, sizes), group=group)
    layout = next((d for d, _ in every if d), None)
    if layout is None:
        return {} if rank == dst else None
    # every rank sees every rank's description, so a bad shard raises on ALL ranks before any point-to-point
    # operation is posted (a one-sided error would leave the others blocked in their sends / receives)
    for r, (d, sz) in enumerate(every):
        if d and d != layout:
            raise ValueError(f"rank {r} holds {d}, rank layout agreed on is {layout}")
        if d and sz != [spans[r][1] - spans[r][0]]:
            raise ValueError(f"rank {r}: local batch sizes {sz} != shard size {spans[r][1] - spans[r][0]}")
    some = next(iter(local.values())) if local else None
    device = some.device if some is not None else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    out: Dict[str, torch.Tensor] = {}
    ops, keep = [], []
    for key, trailing, dtype_name in layout:
        dtype = getattr(torch, dtype_name.replace("torch.", ""))
        x = local.get(key)
        if rank == dst:
            full = torch.empty((total_batch,) + trailing, dtype=dtype, device=device)
            out[key] = full
            if hi > lo:
                full[lo:hi].copy_(x)
            for r, (a, b) in enumerate(spans):
                if r != dst and b > a:
                    view = full[a:b]   # contiguous: the slice is along the leading axis
                    ops.append(dist.P2POp(dist.irecv, torch.view_as_real(view) if view.is_complex() else view,
                                          _global_rank(group, r), group=group))
        elif hi > lo:
            xs = x.contiguous()
            keep.append(xs)
            ops.append(dist.P2POp(dist.isend, torch.view_as_real(xs) if xs.is_complex() else xs,
                                  _global_rank(group, dst), group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out if rank == dst else None


def sharded_trajectory(equation, w0_local: torch.Tensor, total_batch: int, dt: float, num_steps: int,
                       record_every_steps: int = 1, dtype: torch.dtype = torch.complex64, dst: int = 0,
                       group: Optional[dist.ProcessGroup] = None):
    """Run ``get_trajectory_imex`` on this rank's shard (records stay on the device) and hand them to ``dst``."""
    from .solvers import get_trajectory_imex

    local = get_trajectory_imex(equation, w0_local, dt, num_steps=num_steps, record_every_steps=record_every_steps,
                                dtype=dtype, to_cpu=False) if w0_local.shape[0] > 0 else {}
    full = gather_trajectory(local, total_batch, dst=dst, group=group, keys=TRAJECTORY_FIELDS)
    if full is not None and dist.is_initialized() and dist.get_backend(group) != "gloo":
        full = {k: v.cpu() for k, v in full.items()}
    return full


# ----------------------------------------------------------------------------- per-record hand-over, overlapped
def batch_layout(total: int, world_size: int, batch_size: int) -> List[List[Tuple[int, int]]]:
    """Per rank, the ``(start, count)`` batches its contiguous shard of ``total`` samples is generated in (the serial
    batch loop of fno/data_gen/data_gen_McWilliams2d.py:126-152, cut across ranks).  Every rank computes the whole
    table, so the receiver knows what each peer is going to send without asking."""
    if batch_size < 1:
        raise ValueError("batch_size must be positive")
    table = []
    for r in range(world_size):
        lo, hi = shard_batch(total, r, world_size)
        table.append([(s, min(batch_size, hi - s)) for s in range(lo, hi, batch_size)])
    return table


def _page_aligned_empty(shape, dtype: torch.dtype) -> torch.Tensor:
    """Uninitialised CPU tensor whose first byte sits on a page boundary (a view into a slightly larger byte buffer)."""
    import mmap

    nbytes = int(torch.tensor([], dtype=dtype).element_size())
    for d in shape:
        nbytes *= int(d)
    raw = torch.empty(nbytes + mmap.PAGESIZE, dtype=torch.uint8)
    off = (-raw.data_ptr()) % mmap.PAGESIZE
    return raw[off:off + nbytes].view(dtype).view(*shape)


class RecordHandover:
    """Hands the post-processed records of an ensemble job to the host of rank ``dst`` WHILE the steps go on.

    A record of a batch is ``(count, F, *trailing)`` real values on the device (F fields).  ``push`` is called right
    after the record was produced on the current stream and returns at once:

      * on a peer, the record goes to ``dst`` with one point-to-point send (RCCL over xGMI; the collective library
        runs it on its own stream, ordered after the current one);
      * on ``dst``, one receive per peer is posted for that peer's next record (all peers in one group call: seven
        links at once), and everything that has arrived -- and ``dst``'s own record -- is copied into its
        ``(sample, record)`` slot of the page-locked host result by pitched copies on a side stream
        (``tcfd_copy_rows_to_host``), so PCIe runs under the next ``record_every`` steps too.

    ``finish`` posts the receives still outstanding, waits, and returns ``{field: (total, n_rec, *trailing)}`` host
    tensors on ``dst`` (``None`` elsewhere).  What a job pays at the end is the hand-over of the LAST record only.
    No process group: the same pipeline with the device -> host stage alone.  CPU tensors (gloo, the tests): plain
    blocking copies.  Device tensors over a gloo group (``staged``): gloo moves host memory only, so a peer's record goes
    device -> page-locked staging buffer (side stream, event) -> gloo -> ``dst``'s host result, while ``dst``'s own records
    keep the side-stream pitched copies -- the form a one-GPU box can run with two ranks (RCCL refuses two ranks on one
    device: "Duplicate GPU detected"), exercising the stream / event ordering and the slicing of the page-locked result.

    ``mode`` (default: the environment's ``TCFD_HANDOVER``, else ``"p2p"``) picks how a record travels between ranks:

      * ``"p2p"``     a peer and ``dst`` meet in batched point-to-point calls that only the two of them (and the other peers
                      with a record at that moment) post -- point-to-point traffic among a SUBSET of the group's ranks;
      * ``"uniform"`` record interval k is ONE ``batch_isend_irecv`` on EVERY rank of the group: a peer sends its k-th record,
                      or a one-element dummy when it has none left; ``dst`` receives from every peer;
      * ``"gather"``  (alias ``"collective"``) record interval k is ONE ``dist.gather`` to ``dst`` on every rank, records padded
                      to the interval's largest batch, dummies from ranks without a record.

    The last two are the fall-backs for a collective library that rejects subset point-to-point traffic on a group
    communicator: every rank makes the same sequence of group calls (intervals beyond a rank's own records run in
    ``finish``; intervals in which no peer has a record are skipped by everyone).  All three fill the same host result bit
    for bit (tests/test_distributed_cpu.py)."""

    def __init__(self, fields: Sequence[str], total: int, n_rec: int, trailing: Tuple[int, ...], dtype: torch.dtype,
                 layout: List[List[Tuple[int, int]]], device, dst: int = 0, group: Optional[dist.ProcessGroup] = None,
                 lazy_host: bool = True, mode: Optional[str] = None):
        import os

        mode = (mode or os.environ.get("TCFD_HANDOVER") or "p2p").lower()
        mode = {"collective": "gather"}.get(mode, mode)
        if mode not in ("p2p", "uniform", "gather"):
            raise ValueError(f"hand-over mode {mode!r}: expected p2p, uniform or gather (collective)")
        self.mode = mode
        self.fields, self.total, self.n_rec = tuple(fields), total, n_rec
        self.trailing, self.dtype = tuple(trailing), dtype
        self.device = torch.device(device)
        self.group, self.dst = group, dst
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        if len(layout) != self.world:
            raise ValueError(f"layout describes {len(layout)} ranks, the group has {self.world}")
        self.items = [[(s, c, j) for (s, c) in batches for j in range(n_rec)] for batches in layout]
        self.cursor = [0] * self.world
        self._interval = 0             # uniform / gather: the next record interval this rank takes part in
        self._intervals = max((len(it) for it in self.items), default=0)
        self.on_gpu = self.device.type == "cuda"
        self.row_bytes = int(torch.tensor([], dtype=dtype).element_size())
        for d in self.trailing:
            self.row_bytes *= d
        self._keep: list = []          # device buffers a copy or a send still reads
        self._dummies: list = []       # uniform mode, dst: the one-element receive targets of peers without a record
        self._sends: list = []
        self.staged = bool(self.distributed and self.on_gpu and dist.get_backend(group) == "gloo")
        if self.staged or not self.distributed or self.world == 1:
            self.mode = "p2p"          # the staged form and a single process have no group calls to make uniform
        self._outbox: list = []        # staged peers: (event, page-locked copy) of records not yet handed to gloo
        self._inbox: list = []         # staged dst: (work, staging tensor, start, count, record) of receives in flight
        self.host: Optional[Dict[str, torch.Tensor]] = None
        import time

        self.trace = [] if os.environ.get("TCFD_HANDOVER_TRACE") == "1" else None
        self._t0 = time.perf_counter()
        self._host_ready: Dict[str, object] = {}
        self._alloc_thread = None
        self._alloc_pending = None
        self._deferred: list = []      # (buffer, field, ...) copies waiting for their page-locked host field
        self._alloc_error: Optional[BaseException] = None
        if self.rank == dst:
            shape = (total, n_rec) + self.trailing
            if self.on_gpu and lazy_host:
                # Page-locking the result is host work proportional to the WHOLE data set (5.4 GB at BASELINE config 4:
                # ~0.4 s) that only rank dst does -- serial time no number of GPUs shrinks -- so it runs on a helper thread
                # under the steps.  Not as one page-locked allocation: that holds the runtime's memory lock for its whole
                # duration, and every hipMalloc of the stepping thread (the caching allocator asking for a block) waits behind
                # it (measured: stalls of 0.14 - 0.28 s, the 8-GPU job's rank 0 at 1.4 s instead of 1.05).  The result is
                # ordinary memory (allocated at once); the helper page-locks it REGION BY REGION -- the rows of one batch in
                # one field, the unit a record's pitched copy writes into -- in the order the records will arrive, with a
                # pause between regions that lets other threads at the lock; a copy waits (deferred, never blocking the
                # caller) only for its own region.
                import threading

                # page-aligned fields: when a region (the rows of one batch in one field) is a whole number of pages -- every
                # production shape: 64 x 10 x 256 x 256 floats -- each lock covers exactly its region and no pitched copy ever
                # starts inside one registration and runs into the next; other shapes fall back to the edge-page rule below
                self.host = {f: _page_aligned_empty(shape, dtype) for f in self.fields}
                order = []          # (field, start, count): dst's own batches first, then batch k of every peer
                ranks = [dst] + [r for r in range(self.world) if r != dst]
                for k in range(max((len(bt) for bt in layout), default=0)):
                    for r in ranks:
                        if k < len(layout[r]) and layout[r][k][1] > 0:
                            order.extend((f, layout[r][k][0], layout[r][k][1]) for f in self.fields)
                self._host_ready = {(f, s0): threading.Event() for f, s0, _ in order}
                self._registered: list = []
                self._lock_warned = False

                self._stop = threading.Event()      # close(): the helper stops before its next region

                def allocate():
                    import time as _time

                    from . import _lib as L

                    lib = L.load()
                    import mmap

                    PAGE = mmap.PAGESIZE
                    try:
                        self._trace("page-lock begin")
                        edge_pages = set()              # first / last page of every range locked so far
                        for f, s0, c in order:
                            if self._stop.is_set():
                                break
                            view = self.host[f][s0:s0 + c]
                            # whole pages, each locked ONCE: regions are disjoint byte ranges, so only their first / last page
                            # can already belong to a neighbour's lock (overlapping registrations are not portable)
                            first = view.data_ptr() // PAGE * PAGE
                            last = (view.data_ptr() + view.numel() * view.element_size() - 1) // PAGE * PAGE
                            if first in edge_pages:
                                first += PAGE
                            if last in edge_pages:
                                last -= PAGE
                            if last < first:                # wholly inside pages its neighbours locked
                                self._host_ready[(f, s0)].set()
                                continue
                            if lib.tcfd_host_register(first, last + PAGE - first) == 0:
                                self._registered.append(first)
                                edge_pages.add(first)
                                edge_pages.add(last)
                            elif not self._lock_warned:
                                # (a locked-memory limit, say): the copies into this region still work -- through the
                                # runtime's own staging buffers, blocking the side stream's host thread instead of overlapping
                                import warnings

                                self._lock_warned = True
                                warnings.warn("torch-cfd_amd: could not page-lock the ensemble result ("
                                              + lib.tcfd_last_error().decode() + "); records are copied without overlap")
                            self._host_ready[(f, s0)].set()
                            _time.sleep(0.0005)
                        self._trace("page-lock end")
                    except BaseException as e:   # surfaced by the next _land / finish on the caller's thread
                        self._alloc_error = e
                        for ev in self._host_ready.values():
                            ev.set()

                self._alloc_pending = allocate       # started by start_allocation()
            else:
                self.host = {f: torch.empty(shape, dtype=dtype, pin_memory=self.on_gpu) for f in self.fields}
        if self.on_gpu:
            from . import _lib

            self._lib = _lib
            self._clib = _lib.load()
            self.side = torch.cuda.Stream(device=self.device)
        if self.distributed and self.world > 1 and self.on_gpu and not self.staged:
            # ProcessGroupNCCL: batched point-to-point calls among a SUBSET of a group's ranks (a peer and dst) are only
            # defined once the group has run a collective with every rank in it (torch.distributed.batch_isend_irecv) --
            # every rank constructs its hand-over, so this is that collective if the caller has not made one yet
            dist.all_reduce(torch.zeros(1, device=self.device), group=group)

    # -- dst side
    def _land(self, buf: torch.Tensor, start: int, count: int, rec: int, work=None):
        """Copy ``buf`` (count, F, *trailing) into host[f][start:start+count, rec] for every field."""
        if not self.on_gpu:
            if work is not None:
                work.wait()
            for f, name in enumerate(self.fields):
                self.host[name][start:start + count, rec].copy_(buf[:, f])
            return
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.side):
            if work is not None:
                work.wait()                      # orders the side stream behind the receive, the host does not block
            else:
                self.side.wait_stream(main)      # own record: produced on the caller's stream
            for f, name in enumerate(self.fields):
                if self._field_ready(name, start):
                    self._copy_field(buf, f, name, start, count, rec)
                else:                            # its page-locked field is still being allocated: the copy follows later,
                    self._deferred.append((buf, f, name, start, count, rec))   # the caller's thread never waits for it
            done = torch.cuda.Event()
            done.record(self.side)
        self._keep.append((done, buf))

    def _copy_field(self, buf, f: int, name: str, start: int, count: int, rec: int):
        """Pitched device -> host copy of field f of a record into host[name][start:start+count, rec] (current stream = side)."""
        import ctypes

        h = self.host[name]
        rc = self._clib.tcfd_copy_rows_to_host(
            h[start, rec].data_ptr(), self.n_rec * self.row_bytes, buf[:, f].data_ptr(), len(self.fields) * self.row_bytes,
            self.row_bytes, count, ctypes.c_void_p(self.side.cuda_stream))
        self._lib.check(rc, "tcfd_copy_rows_to_host")

    def _field_ready(self, name: str, start: int) -> bool:
        ev = self._host_ready.get((name, start))
        if ev is None:
            return True
        if self._alloc_thread is None:
            self.start_allocation()
        if ev.is_set() and self._alloc_error is not None:
            raise self._alloc_error
        return ev.is_set()

    def _flush_deferred(self, block: bool):
        """Issue the copies that waited for their page-locked field (all of them when ``block``)."""
        if not self._deferred:
            return
        left, touched = [], []
        with torch.cuda.stream(self.side):
            for item in self._deferred:
                buf, f, name, start, count, rec = item
                if block:
                    self._host_field(name, start)
                if self._field_ready(name, start):
                    self._copy_field(buf, f, name, start, count, rec)
                    touched.append(buf)
                else:
                    left.append(item)
            if touched:
                done = torch.cuda.Event()
                done.record(self.side)
                self._keep.extend((done, b) for b in touched)
        self._deferred = left

    def start_allocation(self):
        """Start page-locking the result on the helper thread (idempotent).  The caller picks the moment: page-locking holds
        the process's memory-map lock for long stretches, so a thread that is first-touching fresh host memory at the same
        time (the seeded CPU noise of the initial conditions) crawls -- measured: the first record of the C4 job 0.27 s late.
        ``generate_mcwilliams_dataset`` starts it once the first batch's initial conditions are on the device; without the
        call the first record's hand-over starts it."""
        if self._alloc_pending is not None and self._alloc_thread is None:
            import threading

            self._alloc_thread = threading.Thread(target=self._alloc_pending, name="tcfd-pinned-result", daemon=True)
            self._alloc_thread.start()

    def _trace(self, what: str):
        """(label, seconds since construction) pairs when TCFD_HANDOVER_TRACE=1 (tests/micro/c4_pinned_overlap.py)."""
        if self.trace is not None:
            import time

            self.trace.append((what, round(time.perf_counter() - self._t0, 4)))

    def _host_field(self, name: str, start: int = 0) -> torch.Tensor:
        ev = self._host_ready.get((name, start))
        if ev is not None:
            self.start_allocation()
            if not ev.is_set():
                self._trace(f"wait {name} begin")
                ev.wait()
                self._trace(f"wait {name} end")
            if self._alloc_error is not None:
                raise self._alloc_error
        return self.host[name]

    def _release_finished(self):
        """Drop the device buffers whose copy (dst) or send (peers) has completed: the footprint stays at the records in
        flight instead of growing with the whole data set (a reference-scale job is thousands of samples x 100 records)."""
        self._keep = [(ev, buf) for ev, buf in self._keep if not ev.query()]
        self._sends = [(works, buf) for works, buf in self._sends if not all(w.is_completed() for w in works)]

    # -- staged form (device tensors, gloo group)
    def _flush_outbox(self, block: bool):
        """Hand the staged copies whose device -> host copy has finished to gloo, in push order."""
        while self._outbox:
            ev, host = self._outbox[0]
            if not ev.query():
                if not block:
                    return
                ev.synchronize()
            self._outbox.pop(0)
            works = dist.batch_isend_irecv([dist.P2POp(dist.isend, host, _global_rank(self.group, self.dst), group=self.group)])
            self._sends.append((works, host))

    def _drain_inbox(self, block: bool):
        """Copy the received staging tensors into their (sample, record) slots of the host result."""
        left = []
        for work, stage, s, c, j in self._inbox:
            if not block and not work.is_completed():
                left.append((work, stage, s, c, j))
                continue
            work.wait()
            for f, name in enumerate(self.fields):
                self.host[name][s:s + c, j].copy_(stage[:, f])
        self._inbox = left

    def _post_receives(self):
        """One receive per peer that still has a record to send (its next one), all in one group call."""
        ops, meta = [], []
        for r in range(self.world):
            if r == self.dst or self.cursor[r] >= len(self.items[r]):
                continue
            s, c, j = self.items[r][self.cursor[r]]
            self.cursor[r] += 1
            shape = (c, len(self.fields)) + self.trailing
            stage = (torch.empty(shape, dtype=self.dtype, pin_memory=True) if self.staged
                     else torch.empty(shape, dtype=self.dtype, device=self.device))
            ops.append(dist.P2POp(dist.irecv, stage, _global_rank(self.group, r), group=self.group))
            meta.append((stage, s, c, j))
        if not ops:
            return False
        works = dist.batch_isend_irecv(ops)
        if self.staged:
            works = list(works) if len(works) == len(meta) else [works[0]] + [_Done()] * (len(meta) - 1)
            self._inbox.extend((w, stage, s, c, j) for w, (stage, s, c, j) in zip(works, meta))
            self._drain_inbox(block=False)
            return True
        if len(works) == len(meta):
            for w, (stage, s, c, j) in zip(works, meta):
                self._land(stage, s, c, j, work=w)
        else:   # one coalesced work object for the whole group
            for i, (stage, s, c, j) in enumerate(meta):
                self._land(stage, s, c, j, work=works[0] if i == 0 else _Done())
        return True

    # -- uniform / gather: one group call per record interval on every rank
    def _run_interval(self, k: int, packed: Optional[torch.Tensor]):
        """Record interval k: every rank of the group makes the same call.  ``packed`` is this rank's k-th record on a peer
        that has one (``dst`` lands its own records itself and passes None, like a peer past its last record)."""
        peers = [r for r in range(self.world) if r != self.dst]
        have = [r for r in peers if k < len(self.items[r])]
        if not have:
            return                                   # nobody has anything to hand over: every rank skips the interval
        where = self.device if self.on_gpu else torch.device("cpu")
        gdst = _global_rank(self.group, self.dst)
        if self.mode == "gather":
            cmax = max(self.items[r][k][1] for r in have)
            shape = (cmax, len(self.fields)) + self.trailing
            if self.rank == self.dst:
                slots = [torch.empty(shape, dtype=self.dtype, device=where) for _ in range(self.world)]
                work = dist.gather(slots[self.dst], slots, dst=gdst, group=self.group, async_op=True)
                first = True
                for r in have:
                    s0, c, j = self.items[r][k]
                    self._land(slots[r][:c], s0, c, j, work=work if first else _Done())
                    first = False
                return
            if packed is not None and packed.shape[0] == cmax:
                mine = packed
            else:                                    # a smaller (last) batch or no record at all: padded / dummy payload
                mine = torch.empty(shape, dtype=self.dtype, device=where)
                if packed is not None:
                    mine[:packed.shape[0]].copy_(packed)
            work = dist.gather(mine, None, dst=gdst, group=self.group, async_op=True)
            self._sends.append(([work], mine))
            return
        # uniform: one batched point-to-point call per rank, a one-element dummy where there is no record
        if self.rank == self.dst:
            ops, meta = [], []
            for r in peers:
                if r in have:
                    s0, c, j = self.items[r][k]
                    stage = torch.empty((c, len(self.fields)) + self.trailing, dtype=self.dtype, device=where)
                    meta.append((stage, s0, c, j))
                else:
                    stage = torch.empty(1, dtype=self.dtype, device=where)
                    meta.append(None)
                    self._dummies.append(stage)
                ops.append(dist.P2POp(dist.irecv, stage, _global_rank(self.group, r), group=self.group))
            works = dist.batch_isend_irecv(ops)
            coalesced = len(works) != len(ops)
            first = True
            for i, m in enumerate(meta):
                if m is None:
                    continue
                w = (works[0] if first else _Done()) if coalesced else works[i]
                first = False
                self._land(m[0], m[1], m[2], m[3], work=w)
            if not coalesced:
                self._sends.extend(([works[i]], ops[i].tensor) for i, m in enumerate(meta) if m is None)
            return
        mine = packed if packed is not None else torch.zeros(1, dtype=self.dtype, device=where)
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend, mine, gdst, group=self.group)])
        self._sends.append((works, mine))

    # -- every rank
    def push(self, start: int, rec: int, packed: torch.Tensor):
        mine = self.items[self.rank]
        k = self.cursor[self.rank]
        if k >= len(mine) or mine[k][0] != start or mine[k][2] != rec:
            raise ValueError(f"rank {self.rank}: record (start {start}, index {rec}) pushed out of order; "
                             f"expected {mine[k] if k < len(mine) else 'nothing'}")
        count = mine[k][1]
        want = (count, len(self.fields)) + self.trailing
        if tuple(packed.shape) != want or packed.dtype != self.dtype or not packed.is_contiguous():
            raise ValueError(f"record must be a contiguous {self.dtype} tensor of shape {want}, got "
                             f"{tuple(packed.shape)} {packed.dtype}")
        self.cursor[self.rank] += 1
        self._trace(f"push {start} {rec}")
        if self.on_gpu:
            self._flush_deferred(block=False)
            self._release_finished()
        if self.mode != "p2p":
            if self.rank == self.dst:
                self._land(packed, start, count, rec)
            self._run_interval(self._interval, None if self.rank == self.dst else packed)
            self._interval += 1
        elif self.rank == self.dst:
            self._land(packed, start, count, rec)
            if self.world > 1:
                self._post_receives()
        elif self.staged:
            main = torch.cuda.current_stream(self.device)
            host = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
            with torch.cuda.stream(self.side):
                self.side.wait_stream(main)          # the record was produced on the caller's stream
                host.copy_(packed, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.side)
            self._keep.append((done, packed))
            self._outbox.append((done, host))
            self._flush_outbox(block=False)
        else:
            # through batch_isend_irecv like the receiving side: ProcessGroupNCCL runs batched point-to-point operations on
            # the group's communicator and unbatched ones on a separate two-rank communicator -- a batched receive and an
            # unbatched send would never meet
            works = dist.batch_isend_irecv([dist.P2POp(dist.isend, packed, _global_rank(self.group, self.dst), group=self.group)])
            self._sends.append((works, packed))

    def finish(self) -> Optional[Dict[str, torch.Tensor]]:
        if self.cursor[self.rank] != len(self.items[self.rank]):
            raise RuntimeError(f"rank {self.rank}: {len(self.items[self.rank]) - self.cursor[self.rank]} records were "
                               "never pushed")
        while self.mode != "p2p" and self._interval < self._intervals:
            self._run_interval(self._interval, None)     # the intervals past this rank's own records
            self._interval += 1
        if self.rank == self.dst:
            while self.mode == "p2p" and self.world > 1 and self._post_receives():
                pass
            self._drain_inbox(block=True)
            for works, _ in self._sends:                 # (uniform mode: the dummy receives)
                for work in works:
                    work.wait()
            self._sends.clear()
            self._dummies.clear()
            if self._alloc_pending is not None:
                self.start_allocation()
                self._alloc_thread.join()
                if self._alloc_error is not None:
                    raise self._alloc_error
                self._flush_deferred(block=True)
            if self.on_gpu:
                self.side.synchronize()
            self._keep.clear()
            registered = getattr(self, "_registered", None)
            if registered:
                # release the page locks behind the caller's back: the thread keeps the result alive until every region is
                # unregistered (unlocking 5 GB is tens of milliseconds nobody has to wait for)
                import threading

                host, ptrs, lib = self.host, list(registered), self._clib
                self._registered = []

                def release(alive=host):        # the default argument holds the tensors until the last region is unlocked
                    for ptr in ptrs:
                        lib.tcfd_host_unregister(ptr)

                threading.Thread(target=release, name="tcfd-unlock-result").start()
            return self.host
        self._flush_outbox(block=True)
        for works, _ in self._sends:
            for work in works:
                work.wait()
        if self.on_gpu:
            torch.cuda.current_stream(self.device).synchronize()
        self._sends.clear()
        return None


    def close(self):
        """Release what a hand-over holds when it is dropped BEFORE ``finish`` (an exception in the stepping loop, a cancelled
        job): stop the page-locking helper and wait for it, wait for the copies in flight on the side stream (they write into
        the host result), unregister every page-locked region.  Idempotent; ``finish`` leaves nothing for it to do."""
        stop = getattr(self, "_stop", None)
        if stop is not None:
            stop.set()
        th = getattr(self, "_alloc_thread", None)
        if th is not None and th.is_alive():
            for ev in getattr(self, "_host_ready", {}).values():
                ev.set()
            th.join()
        if getattr(self, "on_gpu", False) and getattr(self, "side", None) is not None:
            try:
                self.side.synchronize()
            except Exception:
                pass
        self._deferred = []
        self._keep = []
        registered, self._registered = getattr(self, "_registered", None) or [], []
        lib = getattr(self, "_clib", None)
        if lib is not None:
            for ptr in registered:
                lib.tcfd_host_unregister(ptr)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Done:
    """Stand-in for the members of a coalesced group whose single work object was already waited on."""

    def wait(self):
        return True

    def is_completed(self):
        return True
