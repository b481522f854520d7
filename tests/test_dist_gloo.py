"""Multi-rank path on CPU: world_size 2 (and 3) gloo jobs run the product's
row partition + y exchange (spmv_scpa_amd/dist.py) end to end."""
import os
import socket
import subprocess
import sys

import pytest

from spmv_scpa_amd import dist as D

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world, mode, chunks, rows):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "_dist_worker.py"), mode,
             str(chunks), str(rows)], env=env, stdout=subprocess.PIPE,
            stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out)
        assert "rank %d ok" % r in out


@pytest.mark.parametrize("world,mode,chunks", [(2, "allgather", 1),
                                               (2, "p2p", 1),
                                               (2, "p2p", 4),
                                               (3, "p2p", 3),
                                               (2, "staged", 4),
                                               (3, "staged", 5),
                                               (2, "staged", 3),
                                               (2, "shards", 4),
                                               (2, "shards", 1),
                                               # rank 0's kernel pick is
                                               # broadcast and built by all
                                               (2, "pick", 4),
                                               (3, "pick", 2),
                                               # halo rows: < a fragment, a
                                               # whole fragment and a half
                                               # the node's real widths
                                               (4, "staged", 4),
                                               (8, "staged", 4),
                                               # config.rccl / per-rank times
                                               (2, "describe", 1),
                                               (3, "describe", 1),
                                               # nnz-balanced partition: ragged
                                               # fragments
                                               (2, "ragged_p2p", 1),
                                               (3, "ragged_p2p", 3),
                                               (8, "ragged_p2p", 2),
                                               (2, "ragged_bcast", 1),
                                               (3, "ragged_bcast", 1),
                                               (8, "ragged_bcast", 1),
                                               (2, "ragged_padded", 1),
                                               (3, "ragged_padded", 1),
                                               (8, "ragged_padded", 1),
                                               (3, "halo", 64),
                                               (3, "halo", 960),
                                               (2, "halo", 32)])
def test_sharded_spmv_gloo(world, mode, chunks):
    run_world(world, mode, chunks, rows=640)


@pytest.mark.parametrize("mode", ["ragged_p2p", "ragged_bcast",
                                  "ragged_padded"])
def test_ragged_exchange_with_empty_ranges(mode):
    """96 rows = 3 hack blocks over 6 ranks: three ranks own nothing, compute
    nothing and still take part in the exchange (a matrix smaller
    than the node)"""
    port = free_port()
    world = 6
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        # the worker builds M = rows_per_rank * world rows: 16 * 6 = 96 = 3 blocks
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "_dist_worker.py"), mode, "2",
             "16"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
            text=True))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0 and "rank %d ok" % r in out, out


def test_nnz_partition_matches_the_library_and_balances():
    """dist.nnz_row_partition == partition_rows_nnz_aligned (csr.h) on skewed
    row-length profiles; aligned, ascending, no empty range, balanced"""
    import numpy as np
    import spmv_scpa_amd as S
    rng = np.random.default_rng(5)
    cases = []
    for M in (1, 31, 32, 33, 100, 4096, 100_003):
        for prof in ("flat", "kkt", "hub", "powerlaw", "empty"):
            if prof == "flat":
                ln = np.full(M, 7)
            elif prof == "kkt":  # 42 entries above, 15 below (nlpkkt160)
                ln = np.where(np.arange(M) < M // 2, 42, 15)
            elif prof == "hub":
                ln = rng.integers(1, 6, M)
                ln[M // 3] = 20 * M + 5
            elif prof == "powerlaw":
                ln = np.minimum((rng.pareto(1.5, M) + 1).astype(np.int64), M)
            else:
                ln = np.zeros(M, dtype=np.int64)
            cases.append((prof, np.concatenate([[0], np.cumsum(ln)])))
    for prof, irp in cases:
        M = len(irp) - 1
        for world in (1, 2, 3, 4, 8):
            got = D.nnz_row_partition(irp, world)
            lib = S.partition_rows_nnz_aligned(irp.astype(np.int32), world)
            assert got == [int(v) for v in lib], (prof, M, world)
            assert got[0] == 0 and got[-1] == M and len(got) == world + 1
            assert all(b >= a for a, b in zip(got, got[1:]))
            assert all(v % 32 == 0 or v == M for v in got)
            nb = -(-M // 32)
            if nb >= world:  # no empty range
                assert all(b > a for a, b in zip(got, got[1:])), (prof, M)
            per, _ = D.partition_balance(irp, got)
            if prof == "kkt" and M >= 4096 and world > 1:
                # within one 32-row block of the ideal share
                assert max(per) - min(per) <= 2 * 32 * 42, (M, world, per)
                ev, _ = D.partition_balance(irp,
                                            D.even_row_partition(M, world))
                assert max(per) <= max(ev)
    # the synthetic form cuts from the generator's row lengths
    st = S.partition_synth_rows_nnz(S.SYNTH_KKT, 100_000, 100_000, 16, 4096,
                                    42, 8)
    A = S.csr_generate(S.SYNTH_KKT, 100_000, 100_000, 16, 4096, 0, 42)
    IRP, _, _ = S.csr_arrays(A)
    assert [int(v) for v in st] == D.nnz_row_partition(IRP, 8)
    S.csr_free(A)
    with pytest.raises(OSError):
        S.partition_rows_nnz_aligned(np.zeros(2, dtype=np.int32), 0)


def test_partition_helpers():
    assert D.even_row_partition(80_000_000, 8) == [10_000_000 * k
                                                   for k in range(9)]
    p = D.even_row_partition(100, 8)
    assert p == [0, 32, 64, 96, 100, 100, 100, 100, 100]
    assert all(v % 32 == 0 for v in D.even_row_partition(10_000_000, 8)[:-1])
    assert D.chunk_bounds(10_000_000, 4) == [0, 2_500_000, 5_000_000,
                                             7_500_000, 10_000_000]
    b = D.chunk_bounds(1000, 3)
    assert b[0] == 0 and b[-1] == 1000 and all(v % 32 == 0 for v in b[:-1])
    assert D.chunk_bounds(40, 8) == [0, 40]
    assert D.chunk_bounds(96, 8) == [0, 32, 64, 96]
    assert D.chunk_bounds(10, 4) == [0, 10]


def test_nnz_partition_property_fuzz():
    """random row-length profiles, alignments and part counts: the library's
    cut == the Python mirror, boundaries aligned and ascending, no empty
    range while blocks suffice, and no cut of the same shape has a smaller
    largest range by more than one block's weight (prefix-nearest cuts are
    within one block of the ideal share)"""
    import numpy as np
    from hypothesis import given, settings, strategies as st
    import spmv_scpa_amd as S

    @settings(max_examples=200, deadline=None)
    @given(st.integers(1, 3000), st.integers(1, 9), st.sampled_from([1, 8, 32]),
           st.integers(0, 2 ** 31 - 1), st.sampled_from(["flat", "exp", "hub"]))
    def prop(M, world, align, seed, shape):
        rng = np.random.default_rng(seed)
        if shape == "flat":
            ln = rng.integers(0, 50, M)
        elif shape == "exp":
            ln = np.minimum(rng.exponential(20.0, M).astype(np.int64), 5000)
        else:
            ln = rng.integers(0, 8, M)
            ln[rng.integers(0, M)] = 100_000
        irp = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
        got = D.nnz_row_partition(irp, world, align)
        lib = S._lib.partition_rows_nnz_aligned
        arr = np.ascontiguousarray(irp, dtype=np.int32)
        import ctypes as C
        p = lib(arr.ctypes.data_as(C.POINTER(C.c_int)), M, world, align)
        out = [p[k] for k in range(world + 1)]
        S._libc_free(p)
        assert got == out
        assert got[0] == 0 and got[-1] == M
        assert all(b >= a for a, b in zip(got, got[1:]))
        assert all(v % align == 0 or v == M for v in got)
        nb = -(-M // align)
        if nb >= world:
            assert all(b > a for a, b in zip(got, got[1:]))
        # within one block of the ideal share (unless forced by "no empty")
        per = [int(irp[got[k + 1]] - irp[got[k]]) for k in range(world)]
        heaviest_block = max(
            int(irp[min(b * align + align, M)] - irp[b * align])
            for b in range(nb))
        if nb >= 4 * world:
            assert max(per) <= irp[-1] / world + 2 * heaviest_block + 1

    prop()
