"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N gloo job.

Exercises the product's partition + exchange logic (spmv_scpa_amd/dist.py)
on CPU tensors.  The local compute is the CPU oracle -- test infrastructure,
allowed here -- because the product has no CPU compute path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import _oracle as O  # noqa: E402
from spmv_scpa_amd import dist as D  # noqa: E402


def skewed_csr(M):
    """the nlpkkt160 shape in miniature: a dense upper part (ragged rows of
    mean 24) over a light lower part (mean 6), global columns"""
    top = (M // 2) // 32 * 32 + 32
    I1, J1, A1 = O.synth_csr(2, top, M, 24, 300, 42)
    I2, J2, A2 = O.synth_csr(2, M - top, M, 6, 300, 43, row0=top)
    IRP = np.concatenate([I1, I1[-1] + I2[1:]]).astype(np.int32)
    return IRP, np.concatenate([J1, J2]), np.concatenate([A1, A2])


def ragged(rank, world, mode, chunks, M):
    """nnz-balanced partition: ranks own DIFFERENT row counts, the fragments
    of y travel by p2p / bcast / padded (dist.RaggedExchange)"""
    IRP, JA, AS = skewed_csr(M)
    starts = D.nnz_row_partition(IRP, world)
    rows = [starts[r + 1] - starts[r] for r in range(world)]
    assert len(set(rows)) > 1, rows  # really ragged
    tiny = -(-M // 32) < world  # fewer hack blocks than ranks: empty ranges
    if tiny:
        assert rows.count(0) == world - -(-M // 32), rows
    else:
        assert all(v % 32 == 0 for v in starts[:-1]) and min(rows) > 0
        _, bal = D.partition_balance(IRP, starts)
        _, bal_even = D.partition_balance(IRP, D.even_row_partition(M, world))
        assert bal < 1.25 and bal < bal_even, (bal, bal_even)
    row0, mine = starts[rank], rows[rank]
    x = torch.from_numpy(O.synth_x(7, 0, M))
    y = torch.full((M,), float("nan"), dtype=torch.float64)
    sub = np.ascontiguousarray(IRP[row0:row0 + mine + 1] - IRP[row0])
    lo, hi = IRP[row0], IRP[row0 + mine]
    calls = []

    def compute(a, b, out=None):
        assert out is None
        calls.append((a, b))
        s2 = np.ascontiguousarray(sub[a:b + 1] - sub[a])
        y[row0 + a:row0 + b] = torch.from_numpy(O.csr_spmv(
            s2, JA[lo + sub[a]:lo + sub[b]], AS[lo + sub[a]:lo + sub[b]],
            x.numpy()))

    sh = D.ShardedSpmv(None, 0, rank, world, None, x, y, chunks=chunks,
                       mode=mode, compute=compute, starts=starts)
    assert sh.mode == mode and sh.rows == mine and sh.row0 == row0
    want = O.csr_spmv(IRP, JA, AS, x.numpy())
    for it in range(3):
        y.fill_(float("nan"))
        del calls[:]
        sh.step()
        assert np.array_equal(y.numpy(), want), (rank, it)
        if mine == 0:
            assert calls == []  # an empty range computes nothing
            continue
        assert calls[0][0] == 0 and calls[-1][1] == mine
        assert len(calls) == (max(1, min(chunks, mine // 32))
                              if mode == "p2p" else 1)
    y.fill_(float("nan"))
    y[row0:row0 + mine] = torch.from_numpy(want[row0:row0 + mine])
    sh.exchange_only()
    assert np.array_equal(y.numpy(), want), rank
    # the even partition handed over as `starts` is the old, unragged path
    if M % (32 * world) == 0:
        ev = D.even_row_partition(M, world)
        sh2 = D.ShardedSpmv(None, 0, rank, world, M // world, x, y, starts=ev,
                            compute=lambda a, b, out=None: None)
        assert sh2.ragged is None and sh2.mode == "allgather"
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode, chunks, rows_per_rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kind, K, W = 2, 12, 500  # ragged family
    M = N = rows_per_rank * world
    if mode.startswith("ragged_"):
        return ragged(rank, world, mode[len("ragged_"):], chunks, M)
    starts = D.even_row_partition(M, world)
    assert starts[rank + 1] - starts[rank] == rows_per_rank
    row0 = starts[rank]
    # each rank generates ONLY its shard (global columns), as bench.py does
    IRP, JA, AS = O.synth_csr(kind, rows_per_rank, N, K, W, 42, row0=row0)
    x = torch.from_numpy(O.synth_x(7, 0, N))
    y = torch.full((M,), float("nan"), dtype=torch.float64)

    def compute(a, b, out=None):
        sub = IRP[a:b + 1] - IRP[a]
        lo, hi = IRP[a], IRP[b]
        res = torch.from_numpy(
            O.csr_spmv(np.ascontiguousarray(sub), JA[lo:hi], AS[lo:hi],
                       x.numpy()))
        if out is None:
            y[row0 + a:row0 + b] = res
        else:
            out.copy_(res)

    if mode == "describe":
        # what every N > 1 line of bench.py says about the job (config.rccl,
        # per-rank kernel times): the collective calls, on the gloo backend
        import importlib.util
        spec = importlib.util.spec_from_file_location(
            "bench", os.path.join(os.path.dirname(HERE), "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        import spmv_scpa_amd as S
        rccl, per_rank = bench.describe_job(
            S, torch, dist, torch.device("cpu"), rank, world, "gloo",
            [float(rank + 1), float(rank + 1)])
        assert rccl["nranks_joined"] == world and len(rccl["devices"]) == world
        assert rccl["backend"] == "gloo" and rccl["library_links"]
        assert per_rank == [float(r + 1) for r in range(world)]
        print("rank %d ok" % rank)
        dist.destroy_process_group()
        return
    if mode == "pick":
        # bench.py's kernel / schedule / tile-height agreement with a FAKE
        # compute: every rank "tunes" something different, all must end up
        # running rank 0's pick and arrange the same exchange for it
        mine = D.Pick(kernel=4 if rank == 0 else 1 + rank % 2,
                      schedule=("chain", "sweep", None)[rank % 3],
                      tile_rows=(8192, 0, 0)[rank % 3])
        pick = D.agree_on_pick(dist, mine)
        assert (pick.kernel, pick.schedule, pick.tile_rows) == (4, "chain", 8192)
        assert pick.same_build(mine) == (rank % 3 == 0)
        assert D.Pick(4, "sweep", 20448).same_build(D.Pick(4, "sweep", 10208))
        assert not D.Pick(4, "chain", 8192).same_build(D.Pick(4, "chain", 4096))
        # the pick decides the arrangement: chain -> `chunks` logical shards
        # per rank, every rank the same collectives (a rank that kept its own
        # pick would issue a different number of all-gathers and hang)
        calls = []

        def fake(a, b, out=None):
            calls.append((a, b))
            compute(a, b, out)
        L = chunks if pick.schedule == "chain" else 1
        sh = D.ShardedSpmv([object()] * L, pick.kernel, rank, world,
                           rows_per_rank, x, y, chunks=1, compute=fake)
        sh.step()
        fI, fJ, fA = O.synth_csr(kind, M, N, K, W, 42)
        assert np.array_equal(y.numpy(), O.csr_spmv(fI, fJ, fA, x.numpy()))
        assert calls == [(rows_per_rank // L * i, rows_per_rank // L * (i + 1))
                         for i in range(L)]
        dist.barrier()
        dist.destroy_process_group()
        print("rank %d ok" % rank)
        return
    if mode == "halo":
        # `chunks` = halo rows: only rows near another rank's range travel
        H = chunks
        sh = D.ShardedSpmv(None, 0, rank, world, rows_per_rank, x, y,
                           mode="halo", halo_rows=H, compute=compute)
        fI, fJ, fA = O.synth_csr(kind, M, N, K, W, 42)
        want = O.csr_spmv(fI, fJ, fA, x.numpy())
        for it in range(2):
            y.fill_(float("nan"))
            sh.step()
            got = y.numpy()
            lo, hi = max(0, row0 - H), min(M, row0 + rows_per_rank + H)
            assert np.array_equal(got[lo:hi], want[lo:hi]), (rank, it)
            assert np.isnan(got[:lo]).all() and np.isnan(got[hi:]).all()
        y.fill_(float("nan"))
        y[row0:row0 + rows_per_rank] = torch.from_numpy(
            want[row0:row0 + rows_per_rank])
        sh.exchange_only()
        assert np.array_equal(y.numpy()[lo:hi], want[lo:hi]), rank
        dist.barrier()
        dist.destroy_process_group()
        print("rank %d ok" % rank)
        return
    if mode == "shards":
        # `chunks` logical shards per rank (bench.py --strong): the shard is
        # the unit of overlap, whatever `chunks` says
        sh = D.ShardedSpmv([object()] * chunks, 0, rank, world, rows_per_rank,
                           x, y, chunks=1, compute=compute)
        assert sh.mode == ("staged" if chunks > 1 else "allgather")
        assert sh.bounds == [rows_per_rank // chunks * i
                             for i in range(chunks + 1)]
    else:
        sh = D.ShardedSpmv(None, 0, rank, world, rows_per_rank, x, y,
                           chunks=chunks, mode=mode, compute=compute)
    for it in range(3):  # iterate: y of step k feeds nothing here, but the
        y.fill_(float("nan"))  # exchange must complete every time
        sh.step()
        fI, fJ, fA = O.synth_csr(kind, M, N, K, W, 42)
        want = O.csr_spmv(fI, fJ, fA, x.numpy())
        got = y.numpy()
        assert not np.isnan(got).any(), (rank, it)
        assert np.array_equal(got, want), (rank, it)
    sh.exchange_only()  # the collectives alone: y must come out the same
    assert np.array_equal(y.numpy(), want), rank
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
