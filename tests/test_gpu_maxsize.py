"""The largest matrix the host API can describe: its structs count entries in
`int` (reference csr.h:9-10, hll.h), so a matrix holds at most INT32_MAX of
them.  One GPU of 288 GB takes such a matrix whole (26 GB as CSR, another 26 GB
per HLL layout, 26 GB for the blocked copy) -- 6.7 times config 3.

* 67 108 863 rows x 32 entries = INT32_MAX - 31 entries: the last rows' entry
  indices lie within a wavefront's stride of the int32 limit (an index formed
  as `first + lane + stride` in 32 bits wraps there), a wavefront per row is
  2^32 work-items (more than one launch holds), the last hack block is ragged
  (31 rows), the blocked copy's padded slots are indexed beyond 2^31
  (unsigned).  Every kernel of both formats, the blocked path in both of its
  main schedules, the selector.
* 66 000 000 rows x 32 = 2.112e9 entries: the blocked path as well, from both
  sources and with both of its main schedules.

* 2^29 rows x 2..4 entries (1.6e9 entries, x and y of 4.3 GB each): the other
  way to be large -- a wavefront, 16 or 8 lanes per row are more work-items
  than a launch holds, so those kernels walk the rows grid-stride; 16.8M hack
  blocks; 131 072 row tiles x 2048 panels of the blocked copy.

* INT32_MAX columns (x of 17 GB, 8192 panels of the blocked copy), columns
  anywhere: column indices up to the int32 limit.
* a hub row of 268 435 455 entries (INT32_MAX / 8 columns of a 1M-row matrix,
  CSR): 131 072 segments of the stream kernel and of the direct kernels' side
  launch, 262 144 beside the blocked copy, their partial sums added up by one
  wavefront each; as HLL a hack block of 8.6e9 slots (103 GB).

Against rows recomputed from the workload definition by the oracle."""
import errno

import numpy as np
import pytest

import _oracle as O
import spmv_scpa_amd as S

pytestmark = pytest.mark.gpu
TIGHT = 1e-12
K, W = 32, 1 << 20


def needs_a_whole_gpu(test):
    """these cases hold 100-150 GB at a time: on a card that somebody else is
    using too, running out of memory is not a finding about this library"""
    import functools

    @functools.wraps(test)
    def run():
        try:
            test()
        except OSError as e:
            if e.errno != errno.ENOMEM:
                raise
            pytest.skip("device memory ran out: %s" % e)
    return run


def _setup(M, kind=S.SYNTH_RANDOM, K=K, W=W, N=0, row0=0):
    if S.device_info(0)[2] < 200 << 30:
        pytest.skip("needs ~150 GB of device memory")
    N = N or M
    dA = S.CsrDevice.generate(kind, M, N, K, W, row0, 42)   # rows row0 .. + M
    assert kind != S.SYNTH_RANDOM or dA.NZ == M * K
    d_x, d_y = S.DevBuffer(N * 8), S.DevBuffer(M * 8)
    S.dev_fill_synth(d_x.ptr, N, 7)
    rng = np.random.default_rng(5)
    rows = np.unique(np.concatenate([
        [0, 31, 32, M // 3, M - 33, M - 32, M - 1],
        rng.integers(0, M, 1_500),
        rng.integers(M - 100_000, M, 500)]))   # entry indices next to 2^31
    want = np.array([O.synth_row_dot(kind, row0 + M, N, K, W, 0, 42, 7,
                                     row0 + int(g)) for g in rows])

    ref = []  # the first kernel's y: every later one must agree with ALL of it

    def check(handle, kernel, tag):
        S._lib.spmv_dev_memset(d_y.ptr, 0xFF, M * 8, None)
        handle.launch(kernel, d_x.ptr, d_y.ptr)
        S.stream_sync()
        y = d_y.to_numpy(np.float64, M)
        assert np.all(np.isfinite(y)), tag
        err = np.max(np.abs(y[rows] - want[:, 0]) / want[:, 1])
        assert err <= TIGHT, (tag, err)
        if not ref:
            ref.append(y)
        else:
            # (relative to the row's magnitude: a row of 2.7e8 entries is
            # ~1e4 in size, and the summation orders differ)
            assert np.max(np.abs(y - ref[0]) / (1.0 + np.abs(ref[0]))) < 1e-11, tag

    return dA, d_x, d_y, check


@needs_a_whole_gpu
def test_at_the_entry_count_limit():
    M = 67_108_863
    assert M * K == 2 ** 31 - 1 - 31
    dA, d_x, d_y, check = _setup(M)
    for k in (2, 4, 0, 1, 3):              # every CSR kernel of the seam
        check(dA, k, "csr kernel %d" % k)
    # the blocked copy: INT32_MAX - 31 entries + bucket padding = slot indices
    # beyond 2^31 (unsigned in the kernels)
    for sched in ("chain", "sweep"):
        dA.build_panels(0, sched)
        assert dA.panels_info()["entries"] == dA.NZ
        check(dA, S.CSR_KERNEL_PANELS, "csr blocked " + sched)
    best, ms = dA.autotune(d_x.ptr, d_y.ptr)
    check(dA, best, "csr autotuned %d" % best)
    dH = dA.to_hll(True)
    dR = dA.to_hll(False)
    dA.release()
    assert dH.slots == M * K and dH.num_blocks == (M + 31) // 32
    for k in (1, 2):
        check(dH, k, "hll col-major kernel %d" % k)
    for k in (0, 3):
        check(dR, k, "hll row-major kernel %d" % k)
    dR.release()
    best, ms = dH.autotune(d_x.ptr, d_y.ptr)
    check(dH, best, "hll autotuned %d" % best)
    dH.release()
    d_x.free()
    d_y.free()


@needs_a_whole_gpu
def test_blocked_path_at_2e9_entries():
    M = 66_000_000
    dA, d_x, d_y, check = _setup(M)
    check(dA, 2, "csr sub-wave")
    # the measured selector at this size (several blocked candidates, each
    # built from 2.1e9 keys), then whatever it picked
    best, ms = dA.autotune(d_x.ptr, d_y.ptr)
    check(dA, best, "csr autotuned %d" % best)
    if best != S.CSR_KERNEL_PANELS:
        dA.build_panels(0)
    check(dA, S.CSR_KERNEL_PANELS, "csr blocked")
    assert dA.panels_info()["entries"] == dA.NZ
    dH = dA.to_hll(True)
    dA.release()
    dH.build_panels(0, "sweep")            # the headline's schedule
    check(dH, S.HLL_KERNEL_PANELS, "hll blocked")
    assert dH.panels_info()["entries"] == M * K
    dH.release()
    d_x.free()
    d_y.free()


@needs_a_whole_gpu
def test_half_a_billion_rows():
    M = 1 << 29
    dA, d_x, d_y, check = _setup(M, S.SYNTH_RAGGED, 3, 4096)
    assert 2 * M <= dA.NZ <= 4 * M and dA.NZ < 2 ** 31
    for k in (2, 4, 0, 1, 3):
        check(dA, k, "csr kernel %d" % k)
    dA.build_panels(0, "chain")
    info = dA.panels_info()
    assert info["entries"] == dA.NZ and info["tiles"] * info["panels"] > 1 << 27
    check(dA, S.CSR_KERNEL_PANELS, "csr blocked chain")
    dH = dA.to_hll(True)
    dR = dA.to_hll(False)
    dA.release()
    assert dH.num_blocks == M // 32 and dH.slots >= dH.NZ
    for k in (1, 2):
        check(dH, k, "hll col-major kernel %d" % k)
    for k in (0, 3):
        check(dR, k, "hll row-major kernel %d" % k)
    dR.release()
    dH.build_panels(0, "chain")
    check(dH, S.HLL_KERNEL_PANELS, "hll blocked chain")
    dH.release()
    d_x.free()
    d_y.free()


@needs_a_whole_gpu
def test_two_billion_columns():
    M, N = 2_000_000, 2 ** 31 - 1
    dA, d_x, d_y, check = _setup(M, S.SYNTH_RANDOM, 32, 1 << 40, N)
    for k in (2, 4, 0, 1):
        check(dA, k, "csr kernel %d" % k)
    for sched in ("chain", "sweep"):
        dA.build_panels(0, sched)
        assert dA.panels_info()["panels"] == 8192
        check(dA, S.CSR_KERNEL_PANELS, "csr blocked " + sched)
    dH = dA.to_hll(True)
    dA.release()
    for k in (1, 2):
        check(dH, k, "hll col-major kernel %d" % k)
    best, ms = dH.autotune(d_x.ptr, d_y.ptr)
    check(dH, best, "hll autotuned %d" % best)
    dH.release()
    d_x.free()
    d_y.free()


@needs_a_whole_gpu
def test_a_hub_row_of_a_quarter_billion_entries():
    M, N = 1_000_000, 2 ** 31 - 1
    # the family's hub row is global row N / 3: a shard of rows around it,
    # the hub at local row M / 3 (which _setup always probes)
    dA, d_x, d_y, check = _setup(M, S.SYNTH_HUB, 6, 4096, N,
                                 row0=N // 3 - M // 3)
    assert dA.NZ > N // 8
    for k in (4, 2, 0, 1, 3):              # stream segments / k_csr_long_seg
        check(dA, k, "csr kernel %d" % k)
    dA.build_panels(0, "chain")            # the row beside the copy
    assert "long row(s) beside" in dA.panels_describe()
    check(dA, S.CSR_KERNEL_PANELS, "csr blocked chain")
    best, ms = dA.autotune(d_x.ptr, d_y.ptr)
    check(dA, best, "csr autotuned %d" % best)
    # ... and as HLL: the hub's hack block alone is 32 x 268M slots (103 GB,
    # filled a lane per slot); its columns go to k_hll_wide in 512 segments
    # of 524 288; the blocked copy cannot index 8.6e9 source slots
    dH = dA.to_hll(True)
    dA.release()
    assert dH.slots > 32 * (N // 8)
    for k in (1, 2):
        check(dH, k, "hll col-major kernel %d" % k)
    with pytest.raises(OSError) as ei:
        dH.build_panels(0)
    assert ei.value.errno == errno.EOVERFLOW
    dH.release()
    d_x.free()
    d_y.free()
