"""Single-process multi-GPU API (spmv_mgpu.h: row shards + grouped
ncclAllGather).  On the 1-GPU box it runs with a world of one device; with
more devices visible it uses all of them."""
import os
import subprocess

import numpy as np
import pytest

import _golden as G
import _oracle as O
import spmv_scpa_amd as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("as_hll", [False, True])
def test_mgpu_host_matrix_matches_oracle(as_hll):
    n = min(S.device_count(), 8)
    M = N = 100_003  # not a multiple of 32 x n: last shard padded
    IRP, JA, AS = O.synth_csr(S.SYNTH_RAGGED, M, N, 24, 5000, 42)
    x = O.synth_x(7, 0, N)
    y_ref = O.csr_spmv(IRP, JA, AS, x)
    scale = O.csr_abs_spmv(IRP, JA, AS, x)
    A = S.csr_from_arrays("mg", M, N, IRP, JA, AS)
    g = S.MultiGpu(n)
    g.load_csr(A, as_hll)
    g.set_x(x)
    ms = g.spmv(iters=3)
    assert len(ms) == 3 and np.all(ms > 0)
    for r in range(n):
        y = g.get_y(r)
        assert np.max(np.abs(y - y_ref) / np.maximum(scale, 1e-300)) <= 1e-12
    # measured kernel choice (may be the blocked path), same answer
    k = g.autotune()
    assert k in ((1, 2, S.HLL_KERNEL_PANELS) if as_hll
                 else (1, 2, 4, S.CSR_KERNEL_PANELS))
    g.spmv(kernel=k, iters=2)
    y = g.get_y(n - 1)
    assert np.max(np.abs(y - y_ref) / np.maximum(scale, 1e-300)) <= 1e-12
    g.destroy()
    S.csr_free(A)


def test_mgpu_generated_shards():
    n = min(S.device_count(), 8)
    rows = 64_000
    g = S.MultiGpu(n)
    g.generate(S.SYNTH_RANDOM, rows, 32, 4096, 42, as_hll=True)
    g.fill_x(7)
    g.spmv(iters=2)
    y = g.get_y(n - 1)
    M = rows * n
    for grow in (0, 31, rows - 1, M - 1):
        want, sc = O.synth_row_dot(S.SYNTH_RANDOM, M, M, 32, 4096, 0, 42, 7, grow)
        assert abs(y[grow] - want) <= 1e-12 * sc
    g.destroy()
    with pytest.raises(OSError):
        S.MultiGpu(S.device_count() + 1)


def test_mgpu_handle_can_be_reloaded_and_leaves_the_current_device_alone():
    """a second load / generate on one handle frees the previous shards
    (device memory stays flat over 10 reloads) and no entry point changes
    the caller's current device (mgpu.hip)"""
    n = min(S.device_count(), 8)
    dev0 = S._lib.spmv_get_device()
    g = S.MultiGpu(n)
    free = []
    for it in range(10):
        g.generate(S.SYNTH_RANDOM, 320_000, 32, 4096, 42, as_hll=bool(it & 1))
        g.fill_x(7)
        g.spmv(iters=1)
        assert S._lib.spmv_get_device() == dev0
        S.stream_sync()
        free.append(S.dev_mem_info()[0])
    # a shard is 123 MB: leaking one per reload would cost 740 MB between
    # reloads 3 and 9 (same format: both odd); the allocator's own caching
    # moves free memory by ~100 MB either way
    assert free[9] > free[3] - (256 << 20), free
    y = g.get_y(0)
    want, sc = O.synth_row_dot(S.SYNTH_RANDOM, 320_000 * n, 320_000 * n, 32,
                               4096, 0, 42, 7, 12345)
    assert abs(y[12345] - want) <= 1e-12 * sc
    g.destroy()
    assert S._lib.spmv_get_device() == dev0


def test_driver_multi_gpu_flag(tmp_path):
    drv = os.path.join(S.ROOT, "spmv_scpa_amd", "bin", "spmv_scpa_amd")
    env = dict(os.environ, OMP_NUM_THREADS="4", SPMV_FORCE_MGPU="1")
    r = subprocess.run([drv, "-m", G.mtx_path("sym70"), "-o", str(tmp_path),
                        "-d", "--iters", "3", "--no-cpu", "-g", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "all-gather" in r.stdout


def test_native_leg_of_bench_prints_the_same_line_shape():
    """bench.py --native-mgpu: the library's own multi-GPU entry points
    (spmv_mgpu_run: shard kernels + grouped ncclAllGather) measured in the
    bench shape, printed with the keys of the torch.distributed line."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run(
        [sys.executable, os.path.join(S.ROOT, "bench.py"), "--native-mgpu",
         "--gpus", "1", "--rows-per-gpu", "640000", "--steps", "3",
         "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
        capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup",
                "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "strong_speedup"):
        assert key in j, key
    c = j["config"]
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["value"] > 0
    assert c["backend"].startswith("native")
    assert c["rccl"]["nranks_joined"] == 1 and len(c["rccl"]["devices"]) == 1
    assert ":" in c["rccl"]["devices"][0] and c["rccl"]["version"]
    assert c["nnz_global"] == 640000 * 32 and j["rows_checked"] >= 258
    assert len(j["roofline"]["kernel_ms_per_rank"]) == 1
    assert j["roofline"]["kernel_ms_avg"] <= j["ms_per_step"] * 1.05


def test_native_path_and_dist_path_give_the_same_y_on_one_shard():
    """the two N > 1 implementations (mgpu.hip and spmv_scpa_amd/dist.py) run
    the same shard with the same kernel: their y must agree bit for bit"""
    from benchlib import devshim as T  # torch's names over the C-ABI
    from spmv_scpa_amd import dist as D
    rows, K, W = 320_000, 32, 1 << 30
    g = S.MultiGpu(1)
    g.generate(S.SYNTH_RANDOM, rows, K, W, 42, as_hll=True)
    g.fill_x(7)
    wall, kms = g.run(1, 1, 2)  # thread-per-row col-major kernel
    assert wall > 0 and len(kms) == 1 and kms[0] > 0
    assert g.comm_ranks() == 1 and len(g.bus_ids()) == 1
    stored, alg, layout = g.shard_info(0)
    assert stored == rows * K and alg > 12 * stored and layout == ""
    y_native = g.get_y(0)
    raw = g.h
    g.destroy()
    # a destroyed handle is refused (-EBADF), not dereferenced
    assert S._lib.spmv_mgpu_fill_x(raw, 7) == -9

    x, y = T.empty(rows), T.zeros(rows)
    S.dev_fill_synth(x.data_ptr(), rows, 7, 0,
                     T.cuda.current_stream().cuda_stream)
    dA = S.CsrDevice.generate(S.SYNTH_RANDOM, rows, rows, K, W, 0, 42)
    dH = dA.to_hll(True)
    dA.release()
    sh = D.ShardedSpmv(dH, 1, 0, 1, rows, x, y, backend=T)
    sh.step()
    T.cuda.synchronize()
    y_dist = y.cpu().numpy()
    dH.release()
    assert np.array_equal(y_native.view(np.uint64), y_dist.view(np.uint64))
    for grow in (0, 12345, rows - 1):
        want, sc = O.synth_row_dot(S.SYNTH_RANDOM, rows, rows, K, W, 0, 42, 7,
                                   grow)
        assert abs(y_native[grow] - want) <= 1e-12 * sc


@pytest.mark.parametrize("as_hll,kernel", [(True, 1), (True, 2), (False, 2)])
@pytest.mark.parametrize("chunks", [2, 4, 5])
def test_native_staged_exchange_runs_on_one_gpu(as_hll, kernel, chunks):
    """spmv_mgpu_set_exchange(chunks, force): the overlapped pipeline of the
    native path -- chunk kernels writing into the chunk-major staging buffer,
    the (here 1-rank) ncclAllGather of chunk c on a second stream under the
    kernel of chunk c+1, the strided copy back -- executed on the 1-GPU box,
    several steps in a row (the staging buffer is reused), against the oracle.
    5 chunks do not divide the rows into whole hack blocks: the launch must
    fall back to the in-place all-gather and still be right."""
    n = min(S.device_count(), 8)
    rows = 64_000                    # per device; 64 000 / (4 x 32) = 500
    g = S.MultiGpu(n)
    g.generate(S.SYNTH_RAGGED, rows, 24, 4096, 42, as_hll=as_hll)
    g.fill_x(7)
    g.set_exchange(chunks, force=True)
    ms = g.spmv(kernel=kernel, warmup=1, iters=3)
    assert len(ms) == 3 and np.all(ms > 0)
    M = rows * n
    y = g.get_y(n - 1)
    probe = [0, 31, 32, rows // chunks - 1, rows // chunks, rows - 1, M - 1]
    for grow in probe:
        want, sc = O.synth_row_dot(S.SYNTH_RAGGED, M, M, 24, 4096, 0, 42, 7, grow)
        assert abs(y[grow] - want) <= 1e-12 * sc, (grow, y[grow], want)
    # the bench shape goes through the same pipeline
    wall, kms = g.run(kernel, 1, 3)
    assert wall > 0 and np.all(kms > 0)
    y2 = g.get_y(0)
    assert np.array_equal(y, y2)
    # a reload keeps the exchange setting and re-sizes the staging buffer
    g.generate(S.SYNTH_RANDOM, 32_000, 16, 2048, 42, as_hll=as_hll)
    g.fill_x(7)
    g.spmv(kernel=kernel, warmup=0, iters=2)
    y3 = g.get_y(0)
    want, sc = O.synth_row_dot(S.SYNTH_RANDOM, 32_000 * n, 32_000 * n, 16, 2048,
                               0, 42, 7, 12_345)
    assert abs(y3[12_345] - want) <= 1e-12 * sc
    with pytest.raises(OSError):
        g.set_exchange(17)
    g.destroy()


@pytest.mark.parametrize("kind,K", [(S.SYNTH_HUB, 6), (S.SYNTH_POWERLAW, 3)],
                         ids=["hub", "powerlaw"])
@pytest.mark.parametrize("as_hll,kernel", [(True, 1), (True, 2), (False, 2),
                                           (False, 0), (False, 4)])
def test_native_chunked_launches_with_long_rows(kind, K, as_hll, kernel):
    """The chunk launches of the overlapped exchange on the reference's
    irregular classes: a hub row as long as the matrix is wide (64 000 entries:
    segments of the CSR side launch, a wide hack block of the HLL one) must
    be summed exactly once -- by the chunk that owns it, whose side launch
    skips the other chunks' rows / blocks -- and land in the staging buffer at
    the right place.  Whole y against the oracle's rows, two steps in a row
    (arrival counters re-armed)."""
    n = min(S.device_count(), 8)
    rows = 64_000
    g = S.MultiGpu(n)
    g.generate(kind, rows, K, 4096, 42, as_hll=as_hll)
    g.fill_x(7)
    g.set_exchange(4, force=True)
    g.spmv(kernel=kernel, warmup=1, iters=2)
    M = rows * n
    y = g.get_y(0)
    hub = M // 3  # SYNTH_HUB's long row (include/spmv_synth.h)
    probe = sorted({0, 31, 32, rows // 4 - 1, rows // 4, hub - 1, hub, hub + 1,
                    rows - 1, M - 1} | set(
        np.random.default_rng(3).integers(0, M, 300).tolist()))
    for grow in probe:
        want, sc = O.synth_row_dot(kind, M, M, K, 4096, 0, 42, 7, grow)
        assert abs(y[grow] - want) <= 1e-12 * sc, (grow, y[grow], want)
    # in-place exchange (chunks = 1): the same y, bit for bit on the direct
    # kernels (their long-row reductions have a fixed order)
    g.set_exchange(1, force=True)
    g.spmv(kernel=kernel, warmup=0, iters=1)
    assert np.array_equal(g.get_y(0), y)
    g.destroy()


@pytest.mark.parametrize("M", [128_000, 100_003])
def test_native_staged_exchange_on_a_host_matrix(M):
    """load_csr + set_exchange(4, force): staged when the rows split into
    chunks of whole hack blocks on every device (128 000), the in-place
    all-gather after the kernel otherwise (100 003: last shard padded)"""
    n = min(S.device_count(), 8)
    IRP, JA, AS = O.synth_csr(S.SYNTH_RAGGED, M, M, 24, 5000, 42)
    x = O.synth_x(7, 0, M)
    y_ref = O.csr_spmv(IRP, JA, AS, x)
    scale = O.csr_abs_spmv(IRP, JA, AS, x)
    A = S.csr_from_arrays("mgx", M, M, IRP, JA, AS)
    for as_hll, kernel in ((True, 1), (False, 2)):
        g = S.MultiGpu(n)
        g.load_csr(A, as_hll)
        g.set_exchange(4, force=True)
        g.set_x(x)
        g.spmv(kernel=kernel, warmup=1, iters=2)
        for r in range(n):
            y = g.get_y(r)
            assert np.max(np.abs(y - y_ref) / np.maximum(scale, 1e-300)) <= 1e-12
        g.destroy()
    S.csr_free(A)


def _skewed_csr(M):
    """the nlpkkt160 shape in miniature: a dense upper part (ragged rows of
    mean 40) over a light lower part (mean 14), global columns"""
    top = (M // 2) // 32 * 32 + 32
    I1, J1, A1 = O.synth_csr(S.SYNTH_RAGGED, top, M, 40, 3000, 42)
    I2, J2, A2 = O.synth_csr(S.SYNTH_RAGGED, M - top, M, 14, 3000, 43, row0=top)
    IRP = np.concatenate([I1, I1[-1] + I2[1:]]).astype(np.int32)
    return IRP, np.concatenate([J1, J2]), np.concatenate([A1, A2])


@pytest.mark.parametrize("nlog", [2, 3, 8])
@pytest.mark.parametrize("as_hll", [False, True])
def test_nnz_partition_on_logical_devices(nlog, as_hll):
    """spmv_mgpu_load_csr_part(_PART_NNZ) on a REHEARSAL handle (nlog logical
    devices on this box's card, copies instead of collectives): the ranges
    are the library's nnz-balanced cut (= dist.nnz_row_partition), ragged,
    32-aligned, entries per device within 10 % of each other where the even
    cut is 2.5x apart; every logical device ends with the whole y, equal to
    the oracle's; the selector's pick runs on the ragged shards too."""
    from spmv_scpa_amd import dist as D
    M = 100_003
    IRP, JA, AS = _skewed_csr(M)
    x = O.synth_x(7, 0, M)
    y_ref = O.csr_spmv(IRP, JA, AS, x)
    scale = O.csr_abs_spmv(IRP, JA, AS, x)
    A = S.csr_from_arrays("skew", M, M, IRP, JA, AS)
    g = S.MultiGpu(nlog, rehearsal=True)
    assert g.comm_ranks() == 0
    g.load_csr(A, as_hll, partition="even")
    st, ent, ragged = g.partition()
    assert not ragged and st == D.even_row_partition(M, nlog)
    assert max(ent) / min(ent) > 2.0
    g.load_csr(A, as_hll, partition="nnz")
    st, ent, ragged = g.partition()
    assert ragged and st == D.nnz_row_partition(IRP, nlog)
    assert sum(ent) == len(JA) and max(ent) / min(ent) < 1.10, ent
    g.set_x(x)
    g.spmv(iters=2)
    for r in range(nlog):
        y = g.get_y(r)
        assert np.max(np.abs(y - y_ref) / np.maximum(scale, 1e-300)) <= 1e-12, r
    k = g.autotune()
    g.spmv(kernel=k, iters=1)
    y = g.get_y(nlog - 1)
    assert np.max(np.abs(y - y_ref) / np.maximum(scale, 1e-300)) <= 1e-12
    wall, kms = g.run(k, 1, 2)
    assert wall > 0 and len(kms) == nlog and np.all(kms > 0)
    g.destroy()
    S.csr_free(A)


def test_nnz_partition_of_a_generated_family_and_empty_ranges():
    """generate(partition="nnz") cuts by the generator's row lengths (no
    matrix on the host); more logical devices than hack blocks leaves the
    trailing ranges empty and the launch skips them"""
    g = S.MultiGpu(4, rehearsal=True)
    rows = 32_000
    M = rows * 4
    g.generate(S.SYNTH_KKT, rows, 16, 4096, 42, as_hll=False, partition="nnz")
    st, ent, ragged = g.partition()
    assert ragged and st[0] == 0 and st[-1] == M
    assert [int(v) for v in S.partition_synth_rows_nnz(
        S.SYNTH_KKT, M, M, 16, 4096, 42, 4)] == st
    assert max(ent) / min(ent) < 1.05
    g.fill_x(7)
    g.spmv(iters=1)
    y = g.get_y(3)
    for grow in (0, st[1] - 1, st[1], st[2], st[3] - 1, M - 1, 77_777):
        want, sc = O.synth_row_dot(S.SYNTH_KKT, M, M, 16, 4096, 0, 42, 7, grow)
        assert abs(y[grow] - want) <= 1e-12 * sc, grow
    g.destroy()
    # 70 rows = 3 hack blocks over 8 logical devices: five empty ranges
    A = S.io_load_csr(G.mtx_path("sym70"))
    IRP, JA, AS = (np.array(v) for v in S.csr_arrays(A))
    x = O.synth_x(7, 0, 70)
    for part in ("even", "nnz"):
        for as_hll in (False, True):
            g = S.MultiGpu(8, rehearsal=True)
            g.load_csr(A, as_hll, partition=part)
            st, ent, _ = g.partition()
            assert st[-1] == 70 and ent.count(0) == 5, (part, st, ent)
            g.set_x(x)
            g.spmv(iters=1)
            assert np.allclose(g.get_y(7), O.csr_spmv(IRP, JA, AS, x),
                               rtol=0, atol=1e-12)
            g.destroy()
    S.csr_free(A)


@pytest.mark.parametrize("xchg", ["bcast", "padded", "p2p"])
def test_ragged_exchange_collectives_run_with_one_rank(xchg):
    """the RCCL side of the ragged exchange on the 1-GPU box: an nnz partition
    over ONE real device is handled as ragged (y holds exactly M rows), and
    with force the exchange runs as 1-rank collectives -- ncclBroadcast in
    place; the padded ncclAllGather through the staging buffer and back; p2p
    has no peer.  y must survive it bit for bit, several steps in a row, and
    the exchange can be timed alone."""
    M = 100_003
    IRP, JA, AS = _skewed_csr(M)
    x = O.synth_x(7, 0, M)
    y_ref = O.csr_spmv(IRP, JA, AS, x)
    scale = O.csr_abs_spmv(IRP, JA, AS, x)
    A = S.csr_from_arrays("skew", M, M, IRP, JA, AS)
    g = S.MultiGpu(1)
    g.set_ragged_exchange(xchg)
    g.load_csr(A, True, partition="nnz")
    st, ent, ragged = g.partition()
    assert ragged and st == [0, M] and ent == [len(JA)]
    g.set_exchange(1, force=True)
    g.set_x(x)
    g.spmv(kernel=1, warmup=1, iters=3)
    y = g.get_y(0)
    assert np.max(np.abs(y - y_ref) / np.maximum(scale, 1e-300)) <= 1e-12
    assert g.exchange_only(5) >= 0.0
    assert np.array_equal(g.get_y(0), y)
    # switching the kind on a loaded handle (re)sizes the staging buffer
    for other in ("padded", "bcast", "p2p"):
        g.set_ragged_exchange(other)
        g.spmv(kernel=2, warmup=0, iters=1)
        assert np.max(np.abs(g.get_y(0) - y_ref)
                      / np.maximum(scale, 1e-300)) <= 1e-12
    with pytest.raises(KeyError):
        g.set_ragged_exchange("ring")
    assert S._lib.spmv_mgpu_set_ragged_exchange(g.h, 7) == -22
    g.destroy()
    S.csr_free(A)


def test_driver_partition_flag(tmp_path):
    """spmv_scpa_amd -g 1 --partition nnz -d: the C driver's option, its
    per-GPU report, and y checked against the serial CSR result"""
    drv = os.path.join(S.ROOT, "spmv_scpa_amd", "bin", "spmv_scpa_amd")
    env = dict(os.environ, OMP_NUM_THREADS="4", SPMV_FORCE_MGPU="1")
    for xchg in ("p2p", "bcast", "padded"):
        r = subprocess.run(
            [drv, "-m", G.mtx_path("sym70"), "-o", str(tmp_path), "-d",
             "--iters", "3", "--no-cpu", "-g", "1", "--partition", "nnz",
             "--ragged-exchange", xchg],
            capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "partition nnz (ragged fragments)" in r.stdout
        assert "GPU 0: rows [0, 70)" in r.stdout
    r = subprocess.run([drv, "-m", G.mtx_path("sym70"), "-o", str(tmp_path),
                        "-g", "1", "--partition", "rows"],
                       capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode != 0 and "even or nnz" in (r.stdout + r.stderr)


@pytest.mark.parametrize("as_hll", [True, False])
@pytest.mark.parametrize("shards,sched", [(2, "sweep"), (4, "chain"),
                                          (4, None)])
def test_logical_shards_overlap_the_blocked_path(as_hll, shards, sched):
    """spmv_mgpu_set_logical_shards: a device's rows as L matrices; the
    staged pipeline then runs with chunk = logical shard -- shard c of every
    device all-gathered on the second stream (here: 1-rank collectives) while
    shard c+1 computes -- for the BLOCKED kernel too, which only runs whole
    matrices (sched None: a direct kernel through the same pipeline).  Whole
    y against the oracle's rows, several steps, the bench shape, shard_info /
    partition summed over the logical shards."""
    n = min(S.device_count(), 8)
    rows = 128_000
    g = S.MultiGpu(n)
    g.set_logical_shards(shards, reserve_cus=8 if sched == "sweep" else 0)
    g.generate(S.SYNTH_RANDOM, rows, 16, 1 << 30, 42, as_hll=as_hll)
    g.fill_x(7)
    g.set_exchange(1, force=True)
    if sched is None:
        kernel = 1 if as_hll else 2
    else:
        kernel = S.HLL_KERNEL_PANELS if as_hll else S.CSR_KERNEL_PANELS
        g.build_panels(sched)
    ms = g.spmv(kernel=kernel, warmup=1, iters=3)
    assert len(ms) == 3 and np.all(ms > 0)
    M = rows * n
    y = g.get_y(n - 1)
    per = rows // shards
    probe = sorted({0, per - 1, per, 2 * per - 1, rows - 1, M - 1} | set(
        np.random.default_rng(5).integers(0, M, 200).tolist()))
    for grow in probe:
        want, sc = O.synth_row_dot(S.SYNTH_RANDOM, M, M, 16, 1 << 30, 0, 42, 7,
                                   grow)
        assert abs(y[grow] - want) <= 1e-12 * sc, (grow, y[grow], want)
    wall, kms = g.run(kernel, 1, 3)
    assert wall > 0 and np.all(kms > 0)
    stored, alg, layout = g.shard_info(0)
    assert stored == rows * 16 and alg > 12 * stored
    st, ent, ragged = g.partition()
    assert not ragged and ent == [rows * 16] * n
    # the selector's pick is propagated to every logical shard (and a sweep
    # pick is rebuilt on a grid that leaves CUs to RCCL)
    k = g.autotune()
    g.spmv(kernel=k, warmup=0, iters=2)
    y2 = g.get_y(0)
    for grow in probe[:40]:
        want, sc = O.synth_row_dot(S.SYNTH_RANDOM, M, M, 16, 1 << 30, 0, 42, 7,
                                   grow)
        assert abs(y2[grow] - want) <= 1e-12 * sc, grow
    g.destroy()


def test_logical_shards_on_logical_devices_and_what_is_refused():
    """rehearsal handle (copies instead of collectives): 3 logical devices x
    2 logical shards through the row-order path; rows that do not split into
    whole hack blocks per shard, or the nnz partition, are refused"""
    g = S.MultiGpu(3, rehearsal=True)
    g.set_logical_shards(2)
    rows = 64_000
    g.generate(S.SYNTH_RAGGED, rows, 24, 4096, 42, as_hll=True)
    g.fill_x(7)
    k = g.autotune()
    g.spmv(kernel=k, warmup=0, iters=2)
    M = rows * 3
    for r in range(3):
        y = g.get_y(r)
        for grow in (0, rows // 2 - 1, rows // 2, rows, M - 1, 99_999):
            want, sc = O.synth_row_dot(S.SYNTH_RAGGED, M, M, 24, 4096, 0, 42,
                                       7, grow)
            assert abs(y[grow] - want) <= 1e-12 * sc, (r, grow)
    with pytest.raises(OSError):  # 64 032 rows: not 2 shards of whole blocks
        g.generate(S.SYNTH_RANDOM, 64_032, 16, 4096, 42)
    with pytest.raises(OSError):
        g.generate(S.SYNTH_RAGGED, rows, 24, 4096, 42, partition="nnz")
    # a load that failed leaves an EMPTY handle, not a half-set one (ADVICE
    # r05: M / N / ranges were set with NULL vectors behind them, and the
    # next step would have handed those to a collective): every step says
    # -EINVAL until something is loaded again
    assert g.partition()[0] == [0, 0, 0, 0]
    for call in (lambda: g.spmv(iters=1), lambda: g.run(1, 0, 1),
                 lambda: g.exchange_only(1), lambda: g.get_y(0),
                 lambda: g.fill_x(7), lambda: g.autotune()):
        with pytest.raises(OSError) as e:
            call()
        assert e.value.errno == 22, e.value
    g.set_logical_shards(1)
    g.generate(S.SYNTH_RAGGED, rows, 24, 4096, 42, partition="nnz")  # fine again
    with pytest.raises(OSError):
        g.set_logical_shards(17)
    g.destroy()


def test_native_bench_times_the_overlapped_arrangements_beside_the_plain_one():
    """bench.py --native-mgpu with an exchange (here forced, one device):
    the PLAIN arrangement is the line (printed provisionally right after its
    K steps); the logical-shard arrangements (RCCL, copy engines) are an
    optional leg timed beside it -- `config.arrangements`, `value_best` --
    and --no-arrangement-choice skips them"""
    import json
    import sys
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    base = [sys.executable, os.path.join(S.ROOT, "bench.py"), "--native-mgpu",
            "--gpus", "1", "--force-exchange", "--rows-per-gpu", "640000",
            "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
            "--no-extras"]
    for extra in (["--window", "0"], ["--window", "65536"]):
        r = subprocess.run(base + extra, capture_output=True, text=True,
                           env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [json.loads(l) for l in r.stdout.splitlines()
                 if l.startswith("{")]
        assert len(lines) == 2 and lines[0]["provisional"] is True
        assert lines[0]["legs_pending"] == ["exchange_alone", "arrangement"]
        j = lines[-1]
        assert j["value"] == lines[0]["value"] > 0 and j["legs_failed"] == []
        c = j["config"]
        assert c["exchange_arrangement"].startswith("plain")
        assert c["logical_shards_per_gpu"] == 1
        arr = c["arrangements"]
        assert arr["plain_ms_per_step"] > 0 and len(arr["alternatives"]) == 2
        assert "RCCL all-gather of shard c" in arr["alternatives"][0]["arrangement"]
        assert "copy engines" in arr["alternatives"][1]["arrangement"]
        assert j["value_best"] >= j["value"] * 0.999
        if arr["winner"] != "plain":
            assert arr["best_ms_per_step"] == j["ms_per_step_best"]
        alt = c["exchange_alternatives_ms"]
        assert set(alt) == {"copy", "allgather"} and min(alt.values()) >= 0
        assert c["exchange_ms_alone"] >= 0
        assert j["rows_checked"] >= 258
    r = subprocess.run(base + ["--window", "0", "--no-arrangement-choice"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["config"]["arrangements"] is None and "value_best" not in j
    assert j["legs_failed"] == []


def test_driver_logical_shards_flag(tmp_path):
    """spmv_scpa_amd -g 1 --logical-shards 2 -d: two matrices per GPU, y
    validated against the serial CSR result; rows that do not split fall
    back to one shard with a warning"""
    drv = os.path.join(S.ROOT, "spmv_scpa_amd", "bin", "spmv_scpa_amd")
    env = dict(os.environ, OMP_NUM_THREADS="4", SPMV_FORCE_MGPU="1")
    for eng in ("rccl", "copy"):
        r = subprocess.run([drv, "-s", "random", "--rows", "64000", "--nnz-row",
                            "16", "--window", "4096", "-o", str(tmp_path), "-d",
                            "--iters", "3", "--no-cpu", "-g", "1",
                            "--logical-shards", "2", "--exchange-engine", eng],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "do not split" not in r.stdout + r.stderr
    r = subprocess.run([drv, "-m", G.mtx_path("sym70"), "-o", str(tmp_path),
                        "-d", "--iters", "2", "--no-cpu", "-g", "1",
                        "--logical-shards", "2"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "do not split into 2 logical shards" in r.stdout + r.stderr


@pytest.mark.parametrize("shards", [1, 4])
@pytest.mark.parametrize("partition", ["even", "nnz"])
def test_copy_engine_exchange(shards, partition):
    """spmv_mgpu_set_exchange_engine("copy"): every device pushes its
    fragment -- or, with logical shards, shard after shard behind the kernels
    -- into its peers' y with peer copies instead of RCCL collectives.  On a
    rehearsal handle (4 logical devices on the card; its only engine) and on a
    real one-device handle with the exchange forced; whole y on every device
    against the oracle, back-to-back steps (the close of a step orders the
    next kernel behind this step's pushes)."""
    if shards > 1 and partition == "nnz":
        pytest.skip("logical shards need the even partition")
    rows = 64_000
    for g, n in ((S.MultiGpu(4, rehearsal=True), 4), (S.MultiGpu(1), 1)):
        g.set_logical_shards(shards)
        if n == 1:
            g.set_exchange_engine("copy")
            g.set_exchange(1, force=True)
        g.generate(S.SYNTH_RAGGED, rows, 24, 4096, 42, as_hll=True,
                   partition=partition)
        g.fill_x(7)
        wall, kms = g.run(1, 1, 6)  # six steps without a sync in between
        assert wall > 0
        M = rows * n
        for r in range(n):
            y = g.get_y(r)
            for grow in (0, rows // 4, rows - 1, M // 2, M - 1):
                want, sc = O.synth_row_dot(S.SYNTH_RAGGED, M, M, 24, 4096, 0,
                                           42, 7, grow)
                assert abs(y[grow] - want) <= 1e-12 * sc, (n, r, grow)
        assert g.exchange_only(3) >= 0.0
        g.destroy()
    with pytest.raises(OSError):  # a rehearsal handle has no communicator
        h = S.MultiGpu(2, rehearsal=True)
        try:
            h.set_exchange_engine("rccl")
        finally:
            h.destroy()


@pytest.mark.parametrize("ndev,shards", [(8, 1), (4, 2), (2, 4)])
def test_config5_whole_80M_problem_on_logical_devices_vs_the_oracle(ndev,
                                                                    shards):
    """BASELINE configs[4] AS A WHOLE: synthetic
    80M x 80M random HLL, hack 32, 32 per row, columns anywhere, even row
    ranges over `ndev` devices (`shards` logical shards of 10M rows each on
    every device), the selector's kernel, one step -- through the library's own
    multi-GPU entry points on a rehearsal handle (ndev LOGICAL devices on this
    box's one card: copies instead of RCCL collectives; every shard, offset,
    launch and fragment of the 8-GPU run exists).  Then y AS EVERY DEVICE
    HOLDS IT after the exchange is compared with the oracle's definition of
    the row (oracle/spmv_oracle.c synth_row_dot) on the first, the last and
    500 random rows of EVERY device's range: 1e-6 relative (north star) and
    1e-12 of the row scale (only the summation order may differ)."""
    import time
    total, K = 80_000_000, 32
    rows, W = total // ndev, 2 * total
    t0 = time.time()
    g = S.MultiGpu(ndev, rehearsal=True)
    if shards > 1:
        g.set_logical_shards(shards, 0)
    g.generate(S.SYNTH_RANDOM, rows, K, W, 42, as_hll=True)
    st, ent, ragged = g.partition()
    assert not ragged and st == [rows * r for r in range(ndev + 1)]
    assert sum(ent) == total * K
    g.fill_x(7)
    k = g.autotune()
    t_setup = time.time() - t0
    ms = g.spmv(k, 0, 1)
    _, _, layout = g.shard_info(0)
    rng = np.random.default_rng(2024 + ndev)
    picks = np.concatenate(
        [np.concatenate([[st[r], st[r + 1] - 1],
                         rng.integers(st[r], st[r + 1], 500)])
         for r in range(ndev)]).astype(np.int64)
    want = np.empty(len(picks))
    scale = np.empty(len(picks))
    for i, grow in enumerate(picks):
        want[i], scale[i] = O.synth_row_dot(S.SYNTH_RANDOM, total, total, K, W,
                                            0, 42, 7, int(grow))
    worst = 0.0
    for r in range(ndev):  # what EVERY device holds after the exchange
        y = g.get_y(r)
        got = y[picks]
        del y
        err12 = np.abs(got - want) / scale
        assert np.max(err12) <= 1e-12, (r, picks[np.argmax(err12)])
        assert np.all(np.abs(got - want) <=
                      1e-6 * np.maximum(np.abs(want), 1e-3 * scale)), r
        worst = max(worst, float(np.max(err12)))
    print("config 5 whole: %d logical devices x %d shard(s), kernel %d (%s), "
          "setup %.1f s, step %.1f ms, %d rows of every range on every "
          "device, worst |dy| / row scale %.2e"
          % (ndev, shards, k, layout, t_setup, float(ms[0]), len(picks),
             worst))
    g.destroy()


def test_kkt_family_nnz_partition_on_8_logical_devices_vs_the_oracle():
    """the nnz-balanced cut (reference csr.c:218-276, 32-aligned) on the
    nlpkkt160-SHAPED family at reduced size over 8 logical devices: ragged
    fragments, every device ends with the whole y, rows of every range
    against the oracle"""
    ndev, rows, K, W = 8, 160_000, 16, 4096
    M = ndev * rows
    g = S.MultiGpu(ndev, rehearsal=True)
    g.generate(S.SYNTH_KKT, rows, K, W, 42, as_hll=True, partition="nnz")
    st, ent, ragged = g.partition()
    assert ragged and st[0] == 0 and st[-1] == M
    assert all(v % 32 == 0 for v in st) and max(ent) / min(ent) < 1.05
    g.fill_x(7)
    k = g.autotune()
    g.spmv(k, 0, 1)
    rng = np.random.default_rng(9)
    picks = np.concatenate(
        [np.concatenate([[st[r], st[r + 1] - 1],
                         rng.integers(st[r], st[r + 1], 200)])
         for r in range(ndev)]).astype(np.int64)
    ws = [O.synth_row_dot(S.SYNTH_KKT, M, M, K, W, 0, 42, 7, int(v))
          for v in picks]
    want = np.array([w for w, _ in ws])
    scale = np.array([s for _, s in ws])
    for r in range(ndev):
        got = g.get_y(r)[picks]
        assert np.max(np.abs(got - want) / np.maximum(scale, 1e-300)) <= 1e-12, r
    g.destroy()


def test_native_leg_keeps_the_childs_provisional_line_when_it_runs_out_of_time():
    """the `native` leg of an N > 1 line is a child process with a timeout cut
    from the run's remaining budget: a child that has printed its (plain
    arrangement) line and then never finishes its own optional legs still
    delivers that line -- flagged provisional, with the timeout it hit"""
    import sys
    sys.path.insert(0, S.ROOT)
    import bench
    from benchlib.native import native_leg
    args = bench.parse_args(["--gpus", "2", "--backend", "gloo", "--steps", "2",
                             "--warmup", "1", "--rows-per-gpu", "320000",
                             "--kernel", "4", "--window", "0"])
    os.environ["SPMV_BENCH_NATIVE_HANG_AFTER_LINE"] = "1"
    try:
        nat = native_leg(args, 2, timeout_s=30)
    finally:
        del os.environ["SPMV_BENCH_NATIVE_HANG_AFTER_LINE"]
    assert nat["provisional"] is True and nat["timed_out_after_s"] == 30
    assert nat["ms_per_step"] > 0 and nat["value"] > 0
    assert nat["backend"].startswith("native REHEARSAL")
    assert len(nat["kernel_ms_per_rank"]) == 2 and nat["rows_checked"] >= 2 * 258
