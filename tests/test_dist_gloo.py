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
                                               (3, "halo", 64),
                                               (3, "halo", 960),
                                               (2, "halo", 32)])
def test_sharded_spmv_gloo(world, mode, chunks):
    run_world(world, mode, chunks, rows=640)


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
