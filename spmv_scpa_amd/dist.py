"""Row-partitioned multi-GPU SpMV: one process per GPU, RCCL over xGMI.

The reference is single-GPU and has no collective; this is new.
Scheme: contiguous row ranges whose boundaries are multiples of
the hack size (32) so no HLL block straddles two GPUs; every rank holds the
full-length x and a full-length y, computes its own fragment
y[row0 : row0 + rows] with the local kernel, then the fragments are
exchanged so that every rank ends with the whole y (the next x of an
iterative method).

Exchange modes
  "allgather"  one in-place all_gather_into_tensor (ncclAllGather) per step.
  "staged"     the shard is cut into k equal row chunks; chunk c of every
               rank is written into a chunk-major staging buffer
               stage[c][rank][:] and all-gathered in place (one
               ncclAllGather per chunk, on RCCL's stream) while the kernel of
               chunk c+1 runs; a final strided copy puts y back into row
               order.  Only collectives, no point-to-point.
  "p2p"        the shard is cut into row chunks; as soon as the kernel of
               chunk c has finished, chunk c is sent to every peer with
               grouped isend/irecv (ncclSend/ncclRecv inside one group,
               landing directly in the peer's y) while the kernel of chunk
               c+1 runs: xGMI is point-to-point, so each of the 7 links
               carries one peer's chunk.
  "halo"       (opt-in, NOT the path BASELINE names) only the rows within
               `halo_rows` of another rank's range travel, point-to-point:
               every rank ends with its fragment plus halo_rows rows either
               side -- all that a following SpMV reads of x when the columns
               of a row stay within halo_rows of the diagonal.  At 8 GPUs the
               full all-gather moves 560 MB into every GPU per step (longer
               than the kernel of a banded matrix); a halo of 2^16 rows moves
               1 MB.

Partitions
  even (default)   equal row counts rounded up to 32 (even_row_partition).
  nnz-balanced     `starts` from nnz_row_partition (the multi-GPU form of the
                   reference's partition_csr_rows, csr.c:218-276; the C side
                   is partition_rows_nnz_aligned, include/csr.h): ranks own
                   DIFFERENT row counts, so the fragments of y are ragged and
                   travel by one of
    "p2p"     grouped isend/irecv of exactly each fragment (default; chunked
              overlap allowed: chunk c of every rank travels under the kernel
              of chunk c+1),
    "bcast"   one broadcast per rank, in place (SURVEY 8e "general case"),
    "padded"  ONE all-gather of fragments padded to the longest into a staging
              buffer + a compaction copy back into row order.

`compute` is pluggable so that the partition + exchange logic is testable on
CPU with the gloo backend (tests/test_dist_gloo.py); the product binds it to
the HIP launch.

Import order: a process that exchanges (world > 1) needs torch.distributed,
and torch must be imported BEFORE spmv_scpa_amd -- the binding then shares the
ROCm runtime torch's wheel has mapped instead of binding the system's (one
runtime per process; spmv_scpa_amd/__init__.py, INTEGRATION.md "Contracts").
With one rank and no exchange, `backend=` takes any module with torch's
`cuda.current_stream()` (benchlib.devshim) and torch is not needed at all.
"""
import numpy as np

HACK = 32


def even_row_partition(total_rows, world, align=HACK):
    """starts[world+1]; equal counts rounded up to `align` (csr.h
    partition_rows_even has the same arithmetic)."""
    per = -(-total_rows // world)
    per = -(-per // align) * align
    return [min(per * k, total_rows) for k in range(world + 1)]


def nnz_row_partition(irp, world, align=HACK):
    """starts[world+1] of `world` contiguous row ranges holding near-equal
    entry counts, boundaries multiples of `align`: cut k is the aligned
    boundary whose prefix entry count is nearest to k/world of the total; no
    range is empty while there are >= world aligned blocks.  Same arithmetic
    as partition_rows_nnz_aligned (include/csr.h, csrc/csr.c) -- the tests
    hold the two against each other.  irp = row offsets of the WHOLE matrix
    (M + 1 entries, any integer dtype)."""
    irp = np.asarray(irp)
    M = len(irp) - 1
    nb = -(-M // align)
    edges = np.minimum(np.arange(nb + 1, dtype=np.int64) * align, M)
    pre = irp[edges].astype(np.int64) - int(irp[0])
    total = int(pre[-1]) if nb else 0
    cut = [0]
    for k in range(1, world):
        if nb < world:
            cut.append(min(k, nb))
            continue
        lo, hi = cut[-1] + 1, nb - (world - k)
        tk = total * k
        idx = int(np.searchsorted(pre * world, tk, side="right")) - 1
        b = max(cut[-1], min(hi, idx))
        c = b
        if b < hi and int(pre[b + 1]) * world - tk < tk - int(pre[b]) * world:
            c = b + 1
        cut.append(min(max(c, lo), hi))
    cut.append(nb)
    return [min(c * align, M) for c in cut]


def partition_balance(irp, starts):
    """(entries per range, max / min over the non-empty ranges)"""
    irp = np.asarray(irp)
    per = [int(irp[starts[k + 1]]) - int(irp[starts[k]])
           for k in range(len(starts) - 1)]
    live = [v for v in per if v > 0]
    return per, (max(live) / min(live) if live else 1.0)


def chunk_bounds(rows, chunks, align=HACK):
    """row chunk boundaries of one shard, multiples of `align`."""
    chunks = max(1, min(chunks, max(1, rows // align)))
    per = -(-rows // chunks)
    per = -(-per // align) * align
    b = [min(per * k, rows) for k in range(chunks + 1)]
    return [v for i, v in enumerate(b) if i == 0 or v > b[i - 1]]


_SCHED_CODE = {None: -1, "steps": 0, "sweep": 1, "chain": 2}
_SCHED_NAME = {v: k for k, v in _SCHED_CODE.items()}


class Pick:
    """What a rank's kernel selector decided: kernel id and, for the 2-D
    blocked path, the schedule ("steps" / "sweep" / "chain") and tile height
    of the copy it built.  The pick decides how the exchange is arranged
    (chunked launches, logical shards, overlap), so all ranks of a job must
    run RANK 0's pick or they would issue different collectives."""

    def __init__(self, kernel, schedule=None, tile_rows=0):
        self.kernel = int(kernel)
        self.schedule = schedule
        self.tile_rows = int(tile_rows or 0)

    def same_build(self, other):
        """would `other`'s blocked copy do for this pick?  (the sweep
        schedule sizes its own tiles: only the schedule must match)"""
        return (self.schedule == other.schedule
                and (self.schedule == "sweep"
                     or self.tile_rows == other.tile_rows))

    def __repr__(self):
        return "Pick(kernel=%d, schedule=%r, tile_rows=%d)" % (
            self.kernel, self.schedule, self.tile_rows)


def agree_on_pick(dist, mine, device=None, group=None):
    """broadcast rank 0's Pick; every rank returns the same object"""
    import torch
    t = torch.tensor([mine.kernel, _SCHED_CODE[mine.schedule], mine.tile_rows],
                     dtype=torch.int64, device=device)
    dist.broadcast(t, 0, group=group)
    k, sc, tr = (int(v) for v in t.tolist())
    return Pick(k, _SCHED_NAME[sc], tr)


class _Done:
    """a collective that has already completed (rehearsal path)"""

    def wait(self):
        return True


def all_gather_fragments(dist, out, mine, group=None):
    """in-place all-gather of equally sized fragments, asynchronous.

    Product path: RCCL (`nccl` backend), one ncclAllGather on the
    communicator's stream.  REHEARSAL path (`gloo` backend with tensors on a
    GPU -- several ranks sharing the one card of a test box, where RCCL
    refuses duplicate devices): gloo has no GPU all-gather, so the fragment is
    staged through host memory, synchronously.  Slow by construction; it
    exists so that every line of the multi-rank control flow (pick agreement,
    arrangement selection, logical shards, cross-rank result check) can run
    before a real node does."""
    if dist.get_backend(group) == "gloo" and mine.is_cuda:
        import torch
        torch.cuda.current_stream().synchronize()
        host = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(host, mine.cpu().contiguous().view(-1),
                                    group=group)
        out.view(-1).copy_(host)
        return _Done()
    return dist.all_gather_into_tensor(out, mine, group=group, async_op=True)


class ShardExchange:
    """Exchange of equally sized y fragments between `world` ranks.

    y is the full-length vector (rows_per_rank * world); rank r owns
    y[r*rows_per_rank : (r+1)*rows_per_rank].
    """

    def __init__(self, y, rank, world, rows_per_rank, mode="allgather",
                 group=None):
        import torch.distributed as dist
        self.dist = dist
        self.y, self.rank, self.world = y, rank, world
        self.rows = rows_per_rank
        self.mode = mode
        self.group = group
        assert y.numel() == rows_per_rank * world
        self.mine = y[rank * rows_per_rank:(rank + 1) * rows_per_rank]

    def gather_all(self, force=False):
        """whole fragments, in place"""
        if self.world == 1 and not force:
            return None
        if self.mode == "allgather":
            return all_gather_fragments(self.dist, self.y, self.mine,
                                        self.group)
        return self.send_chunk(0, self.rows)

    def send_chunk(self, a, b):
        """rows [a, b) of every rank's fragment -> every other rank (grouped
        point-to-point; returns the list of requests)"""
        if self.world == 1 or b <= a:
            return []
        ops = []
        for step in range(1, self.world):
            dst = (self.rank + step) % self.world
            src = (self.rank - step) % self.world
            ops.append(self.dist.P2POp(self.dist.isend, self.mine[a:b], dst,
                                       group=self.group))
            ops.append(self.dist.P2POp(
                self.dist.irecv,
                self.y[src * self.rows + a:src * self.rows + b], src,
                group=self.group))
        return self.dist.batch_isend_irecv(ops)


class RaggedExchange:
    """Exchange of y fragments of DIFFERENT lengths: rank r owns
    y[starts[r] : starts[r+1]] of the full-length y (starts[-1] rows)."""

    def __init__(self, y, rank, world, starts, mode="p2p", group=None):
        import torch.distributed as dist
        if mode not in ("p2p", "bcast", "padded"):
            raise ValueError("ragged fragments travel by p2p, bcast or padded")
        self.dist, self.group = dist, group
        self.y, self.rank, self.world = y, rank, world
        self.starts = [int(v) for v in starts]
        assert len(self.starts) == world + 1 and y.numel() == self.starts[-1]
        self.mode = mode
        self.mine = y[self.starts[rank]:self.starts[rank + 1]]
        self.stage = None
        if mode == "padded":
            import torch
            self.longest = max(self.starts[r + 1] - self.starts[r]
                               for r in range(world))
            self.stage = torch.empty(world, max(self.longest, 1),
                                     dtype=y.dtype, device=y.device)

    def frag(self, r):
        return self.y[self.starts[r]:self.starts[r + 1]]

    def _rehearsal(self):
        """gloo backend with tensors on a GPU (ranks sharing a test box's one
        card): see all_gather_fragments"""
        return self.dist.get_backend(self.group) == "gloo" and self.y.is_cuda

    def _rehearse_rows(self, mine_ab, peer_ab):
        """host-staged stand-in for every mode: one CPU broadcast per rank"""
        import torch
        torch.cuda.current_stream().synchronize()
        for r in range(self.world):
            ab = mine_ab if r == self.rank else peer_ab(r)
            if not ab or ab[1] <= ab[0]:
                continue
            dst = self.y[self.starts[r] + ab[0]:self.starts[r] + ab[1]]
            host = (dst.cpu() if r == self.rank
                    else torch.empty(ab[1] - ab[0], dtype=self.y.dtype))
            self.dist.broadcast(host, r, group=self.group)
            if r != self.rank:
                dst.copy_(host)
        return _Done()

    def send_rows(self, mine_ab, peer_ab):
        """grouped point-to-point: rows mine_ab = (a, b) of MY fragment to
        every peer, rows peer_ab(r) of rank r's fragment from rank r (local
        row numbers; None or an empty range: nothing travels)"""
        if self._rehearsal():
            return self._rehearse_rows(mine_ab, peer_ab)
        ops = []
        for step in range(1, self.world):
            dst = (self.rank + step) % self.world
            src = (self.rank - step) % self.world
            if mine_ab and mine_ab[1] > mine_ab[0]:
                ops.append(self.dist.P2POp(
                    self.dist.isend, self.mine[mine_ab[0]:mine_ab[1]], dst,
                    group=self.group))
            ab = peer_ab(src)
            if ab and ab[1] > ab[0]:
                ops.append(self.dist.P2POp(
                    self.dist.irecv,
                    self.y[self.starts[src] + ab[0]:self.starts[src] + ab[1]],
                    src, group=self.group))
        return self.dist.batch_isend_irecv(ops) if ops else []

    def gather_all(self, force=False):
        """whole fragments; -> work handle(s) for wait_all, then finish()"""
        if self.world == 1 and not force:
            return None
        if self._rehearsal() and self.mode != "padded":
            return self._rehearse_rows(
                (0, self.mine.numel()),
                lambda r: (0, self.starts[r + 1] - self.starts[r]))
        if self.mode == "p2p":
            return self.send_rows(
                (0, self.mine.numel()),
                lambda r: (0, self.starts[r + 1] - self.starts[r]))
        if self.mode == "bcast":
            return [self.dist.broadcast(self.frag(r), r, group=self.group,
                                        async_op=True)
                    for r in range(self.world)
                    if self.starts[r + 1] > self.starts[r]]
        n = self.mine.numel()
        self.stage[self.rank, :n].copy_(self.mine)
        return all_gather_fragments(self.dist, self.stage.view(-1),
                                    self.stage[self.rank], self.group)

    def finish(self):
        """padded mode: staging rows back into row order (the compaction)"""
        if self.mode != "padded":
            return
        for r in range(self.world):
            n = self.starts[r + 1] - self.starts[r]
            if n and r != self.rank:
                self.frag(r).copy_(self.stage[r, :n])


def wait_all(work):
    if work is None:
        return
    if isinstance(work, (list, tuple)):
        for w in work:
            w.wait()
    else:
        work.wait()


class StagedExchange:
    """Chunk-major staging for overlapped all-gathers (mode "staged")."""

    def __init__(self, y, rank, world, rows_per_rank, chunks, group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.y, self.rank, self.world = y, rank, world
        self.rows, self.k = rows_per_rank, chunks
        assert rows_per_rank % chunks == 0
        self.ch = rows_per_rank // chunks
        self.stage = torch.empty(chunks, world, self.ch, dtype=y.dtype,
                                 device=y.device)

    def out(self, c):
        """where this rank's kernel writes chunk c"""
        return self.stage[c, self.rank]

    def gather(self, c):
        return all_gather_fragments(self.dist, self.stage[c].view(-1),
                                    self.stage[c, self.rank], self.group)

    def finish(self):
        """stage[c][r][:] -> y[r*rows + c*ch : ...] (row order)"""
        self.y.view(self.world, self.k, self.ch).copy_(
            self.stage.transpose(0, 1))


def _is_even(starts, rows_per_rank):
    """does `starts` describe equal fragments of rows_per_rank rows?"""
    if rows_per_rank is None:
        return False
    return all(int(starts[k]) == k * rows_per_rank
               for k in range(len(starts)))


class ShardedSpmv:
    """One rank's part of y = A x over `world` GPUs.

    mat            CsrDevice or HllDevice holding rows [row0, row0+rows), or
                   a list of L such handles of rows/L rows each ("logical
                   shards": the fixed 80M-row problem of config 5 is 8 shards
                   of 10M rows, 8/world per GPU, each int32-safe); a logical
                   shard is then the unit of overlap -- shard c is all-gathered
                   while shard c+1 computes -- and `chunks` is ignored
    x, y           full-length torch tensors on this rank's GPU
    chunks         > 1: overlap the exchange of chunk c with the kernel of c+1
    """

    def __init__(self, mat, kernel, rank, world, rows_per_rank, x, y,
                 waves_per_block=0, chunks=1, mode=None, compute=None,
                 force_exchange=False, halo_rows=0, starts=None,
                 backend=None):
        # backend: the module that provides cuda.current_stream() (and the
        # tensors x / y): torch, or -- one rank, no exchange -- any stand-in
        # with the same names (benchlib.devshim: the single-GPU bench never
        # imports torch).  With ranks it is torch: the exchange is
        # torch.distributed's
        if backend is None:
            import torch as backend
        self.torch = backend
        self.mats = list(mat) if isinstance(mat, (list, tuple)) else None
        if self.mats is not None and len(self.mats) == 1:
            mat, self.mats = self.mats[0], None
        self.mat, self.kernel = mat, kernel
        self.rank, self.world, self.rows = rank, world, rows_per_rank
        self.x, self.y = x, y
        self.waves = waves_per_block
        self.row0 = rank * rows_per_rank if rows_per_rank is not None else 0
        self.compute = compute or self._launch
        self.is_hll = hasattr(self.mats[0] if self.mats else mat, "num_blocks")
        self.force_exchange = force_exchange
        self.halo_rows = int(halo_rows)
        self.staged = self.ex = None
        self.ragged = None
        if starts is not None and not _is_even(starts, rows_per_rank):
            self._init_ragged(starts, chunks, mode)
            return
        if self.mats is not None:
            L = len(self.mats)
            if rows_per_rank % L or (rows_per_rank // L) % HACK:
                raise ValueError("logical shards must split the rank's rows "
                                 "evenly at multiples of %d" % HACK)
            self.shard_rows = rows_per_rank // L
            self.bounds = [self.shard_rows * i for i in range(L + 1)]
            mat = self.mats[0]
        else:
            self.bounds = chunk_bounds(rows_per_rank, chunks)
        if mode == "halo" and self.halo_rows <= 0:
            raise ValueError("mode 'halo' needs halo_rows > 0")
        if mode is None:
            mode = "allgather" if len(self.bounds) <= 2 else "staged"
        if mode == "staged":
            k = len(self.bounds) - 1
            ch = rows_per_rank // max(k, 1)
            if (k < 2 or rows_per_rank % k or ch % HACK
                    or self.bounds != [ch * i for i in range(k + 1)]):
                mode, self.bounds = "allgather", [0, rows_per_rank]
        self.mode = mode
        if world == 1 and not force_exchange:
            # nothing travels: no exchange object (and no torch.distributed)
            self.mode = mode if mode in ("allgather", "halo") else "allgather"
            if self.mats is None:
                self.bounds = [0, rows_per_rank]
            return
        self.staged = (StagedExchange(y, rank, world, rows_per_rank,
                                      len(self.bounds) - 1)
                       if mode == "staged" else None)
        self.ex = ShardExchange(y, rank, world, rows_per_rank,
                                "p2p" if mode == "p2p" else "allgather")

    def _init_ragged(self, starts, chunks, mode):
        """ranks own different row counts (nnz-balanced partition)"""
        if self.mats is not None:
            raise ValueError("logical shards need the even partition")
        if mode == "halo":
            raise ValueError("mode 'halo' needs the even partition")
        starts = [int(v) for v in starts]
        if len(starts) != self.world + 1 or starts[0] != 0 or any(
                b < a for a, b in zip(starts, starts[1:])) or any(
                v % HACK and v != starts[-1] for v in starts):
            raise ValueError("starts: world+1 ascending row offsets from 0, "
                             "multiples of %d (or the row count)" % HACK)
        if mode in (None, "allgather", "staged"):
            mode = "p2p"
        self.mode = mode
        self.rows = starts[self.rank + 1] - starts[self.rank]
        self.row0 = starts[self.rank]
        self.ragged = RaggedExchange(self.y, self.rank, self.world, starts,
                                     mode)
        # row chunks of EVERY rank's fragment (p2p only): the peers' bounds
        # follow from starts, so no metadata travels
        k = chunks if mode == "p2p" else 1
        self.peer_bounds = [chunk_bounds(starts[r + 1] - starts[r], k)
                            for r in range(self.world)]
        self.bounds = self.peer_bounds[self.rank]
        self.nsteps = max(len(b) - 1 for b in self.peer_bounds)

    def _ragged_chunk(self, r, c):
        b = self.peer_bounds[r]
        return (b[c], b[c + 1]) if c + 1 < len(b) else None

    def _ragged_step(self, events=None, kernels=True):
        if events and kernels:
            events[0].record()
        exchange = self.world > 1 or self.force_exchange
        pending = []
        last = len(self.bounds) - 2
        for c in range(self.nsteps):
            mine = self._ragged_chunk(self.rank, c)
            if mine and kernels and mine[1] > mine[0]:
                self.compute(mine[0], mine[1])
            if events and kernels and c == max(last, 0):
                events[1].record()
            if not exchange:
                continue
            if self.mode == "p2p":
                pending.append(self.ragged.send_rows(
                    mine, lambda r, c=c: self._ragged_chunk(r, c)))
            else:  # whole fragments, one step
                pending.append(self.ragged.gather_all(self.force_exchange))
        if events and kernels and self.nsteps == 0:
            events[1].record()
        for w in pending:
            wait_all(w)
        self.ragged.finish()

    # product compute: the HIP kernel on torch's current stream; rows [a, b)
    # go to y (row order) or, when `out` is given, to out[0 : b-a]
    def _launch(self, a, b, out=None):
        st = self.torch.cuda.current_stream().cuda_stream
        d_x = self.x.data_ptr()
        d_y = (self.y.data_ptr() + 8 * self.row0 if out is None
               else out.data_ptr() - 8 * a)
        if self.mats is not None:  # one whole logical shard per call
            assert a % self.shard_rows == 0 and b - a == self.shard_rows
            self.mats[a // self.shard_rows].launch(
                self.kernel, d_x, d_y + 8 * a, waves_per_block=self.waves,
                stream=st)
        elif a == 0 and b == self.rows:
            self.mat.launch(self.kernel, d_x, d_y,
                            waves_per_block=self.waves, stream=st)
        elif self.is_hll:
            self.mat.launch(self.kernel, d_x, d_y, waves_per_block=self.waves,
                            stream=st, blocks=(a // HACK, -(-b // HACK)))
        else:
            self.mat.launch(self.kernel, d_x, d_y, waves_per_block=self.waves,
                            stream=st, rows=(a, b))

    def halo_slices(self, other):
        """(send, recv): row ranges (global) this rank sends to / receives
        from rank `other` in mode "halo" -- the part of the sender's fragment
        within halo_rows of the receiver's own rows; None when empty.  What a
        rank holds afterwards is its fragment plus halo_rows rows on either
        side: all a following SpMV needs of x when the columns of a row stay
        within halo_rows of the diagonal."""
        def need(owner, user):
            lo = max(owner * self.rows, user * self.rows - self.halo_rows)
            hi = min((owner + 1) * self.rows,
                     (user + 1) * self.rows + self.halo_rows)
            return (lo, hi) if hi > lo else None
        return need(self.rank, other), need(other, self.rank)

    def halo_exchange(self):
        dist = self.ex.dist
        ops = []
        for other in range(self.world):
            if other == self.rank:
                continue
            send, recv = self.halo_slices(other)
            if send:
                ops.append(dist.P2POp(dist.isend, self.y[send[0]:send[1]],
                                      other, group=self.ex.group))
            if recv:
                ops.append(dist.P2POp(dist.irecv, self.y[recv[0]:recv[1]],
                                      other, group=self.ex.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def exchange_only(self):
        """the collectives of one step without the kernels (what the
        exchange costs when nothing hides it)"""
        if self.world == 1 and not self.force_exchange:
            return
        if self.ragged is not None:
            self._ragged_step(kernels=False)
            return
        if self.mode == "halo":
            wait_all(self.halo_exchange())
            return
        nb = len(self.bounds) - 1
        if self.staged is not None:
            for w in [self.staged.gather(c) for c in range(nb)]:
                wait_all(w)
            self.staged.finish()
            return
        pending = []
        for c in range(nb):
            if nb == 1:
                pending.append(self.ex.gather_all(self.force_exchange))
            else:
                pending.append(self.ex.send_chunk(self.bounds[c],
                                                  self.bounds[c + 1]))
        for w in pending:
            wait_all(w)

    def step(self, events=None):
        """one SpMV (+ exchange).  events = (start, stop) torch events
        recorded around the kernel launches on the current stream."""
        if self.ragged is not None:
            self._ragged_step(events)
            return
        if events:
            events[0].record()
        if self.world == 1 and not self.force_exchange:
            for c in range(len(self.bounds) - 1) if self.mats else (None,):
                if c is None:
                    self.compute(0, self.rows)
                else:
                    self.compute(self.bounds[c], self.bounds[c + 1])
            if events:
                events[1].record()
            return
        pending = []
        nb = len(self.bounds) - 1
        if self.mode == "halo":  # whole fragment, then the boundary rows
            for c in range(nb):
                self.compute(self.bounds[c], self.bounds[c + 1])
            if events:
                events[1].record()
            wait_all(self.halo_exchange())
            return
        if self.staged is not None:
            for c in range(nb):
                self.compute(self.bounds[c], self.bounds[c + 1],
                             self.staged.out(c))
                if c == nb - 1 and events:
                    events[1].record()
                pending.append(self.staged.gather(c))
            for w in pending:
                wait_all(w)
            self.staged.finish()
            return
        for c in range(nb):
            a, b = self.bounds[c], self.bounds[c + 1]
            self.compute(a, b)
            if c == nb - 1 and events:
                events[1].record()
            # torch's NCCL ops wait for the current stream's work enqueued so
            # far (the kernel of this chunk) and run on the communicator's
            # own stream: the next chunk's kernel overlaps with them
            if nb == 1:
                pending.append(self.ex.gather_all(self.force_exchange))
            else:
                pending.append(self.ex.send_chunk(a, b))
        for w in pending:
            wait_all(w)
