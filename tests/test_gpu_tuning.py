"""Tuning expectations of the measured selector and of the blocked copy's
geometry, kept apart from the parity tests: a change of heuristics may turn
these red without saying anything about correctness.
They state what the committed profiles/ and DESIGN.md claim."""
import numpy as np
import pytest

import spmv_scpa_amd as S

pytestmark = pytest.mark.gpu


def test_sweep_geometry_for_config3():
    """10M x 10M, columns anywhere: the sweep schedule holds one row tile per
    workgroup and round -- 2 rounds of 256 tiles of <= 20448 rows with
    2^17-column panels, or of 512 tiles of <= 10208 rows with 2^18."""
    M = N = 10_000_000
    dA = S.CsrDevice.generate(S.SYNTH_RANDOM, M, N, 32, 1 << 30, 0, 42)
    dH = dA.to_hll(True)
    dA.release()
    try:
        dH.build_panels(0, "sweep")
        info = dH.panels_info()
        assert info["steps"] == 1
        assert (info["tiles"], info["panels"]) in ((512, 77), (1022, 39))
        # leaving CUs to a neighbour kernel shrinks the grid, not the coverage
        dH.build_panels(0, "sweep", reserve_cus=16)
        info2 = dH.panels_info()
        assert info2["entries"] == info["entries"] and info2["steps"] == 1
        assert info2["tiles"] >= info["tiles"]
    finally:
        dH.release()


def test_autotune_picks_the_blocked_path_when_columns_are_anywhere():
    """one rank's shard of config 5 (80M columns): ~6 ms direct vs ~3 ms
    blocked -- the selector must see that.  WHICH blocked schedule wins is a
    timing outcome (sweep 3.0 vs chain 3.4 ms here, within noise of each
    other at 10M columns): the property is the pick and its time, not the
    schedule's name."""
    M, N = 10_000_000, 80_000_000
    dA = S.CsrDevice.generate(S.SYNTH_RANDOM, M, N, 32, 1 << 30, 3 * M, 42)
    dH = dA.to_hll(True)
    dA.release()
    d_x, d_y = S.DevBuffer(N * 8), S.DevBuffer(M * 8)
    try:
        S.dev_fill_synth(d_x.ptr, N, 7)
        best, ms = dH.autotune(d_x.ptr, d_y.ptr)
        assert best == S.HLL_KERNEL_PANELS
        assert dH.panels_schedule() in ("sweep", "chain")
        direct = float(np.median(dH.time(1, d_x.ptr, d_y.ptr, 1, 3)))
        assert ms < 0.7 * direct, (ms, direct)
    finally:
        dH.release()
        d_x.free()
        d_y.free()


def test_autotune_times_small_matrices_out_of_the_infinity_cache():
    """config 2 (212 MB working set < 256 MB Infinity Cache): the selector
    flushes between launches, so its time is the HBM-regime time the bench
    reports, within noise of an explicitly flushed timing of its pick."""
    M = N = 1_000_000
    dA = S.CsrDevice.generate(S.SYNTH_BANDED, M, N, 16, 0, 0, 42)
    d_x, d_y = S.DevBuffer(N * 8), S.DevBuffer(M * 8)
    try:
        S.dev_fill_synth(d_x.ptr, N, 7)
        best, ms = dA.autotune(d_x.ptr, d_y.ptr)
        flushed = float(np.median(dA.time(best, d_x.ptr, d_y.ptr, warmup=2,
                                          iters=10, flush_bytes=1 << 30)))
        cached = float(np.median(dA.time(best, d_x.ptr, d_y.ptr, warmup=2,
                                         iters=10, flush_bytes=0)))
        assert abs(ms - flushed) <= 0.25 * flushed + 0.01, (ms, flushed, cached)
    finally:
        dA.release()
        d_x.free()
        d_y.free()


@pytest.mark.parametrize("kind,M,K,W", [
    (S.SYNTH_POWERLAW, 2_000_000, 3, 1 << 30),   # webbase / amazon class
    (S.SYNTH_HUB, 1_000_000, 6, 4096),           # dc1 class
], ids=["powerlaw", "hub"])
def test_selector_on_the_reference_s_irregular_classes(kind, M, K, W):
    """on very short rows the selector MEASURES the
    thread-per-row kernel (the reference's plots show it winning on roadNet /
    amazon, cuda_csr.cu:19-31), and the block-per-row kernel when one row is
    far longer than the rest -- and neither wins on wave64: the pick is the
    stream kernel or the blocked copy, several times faster.  A hub row of
    131 072 entries must not serialise the launch (stream kernel: segments;
    blocked copy: the row beside it)."""
    dA = S.CsrDevice.generate(kind, M, M, K, W, 0, 42)
    d_x, d_y = S.DevBuffer(M * 8), S.DevBuffer(M * 8)
    dH = None
    try:
        S.dev_fill_synth(d_x.ptr, M, 7)
        best, ms = dA.autotune(d_x.ptr, d_y.ptr)
        t = dA.tune_times()
        assert best in (4, S.CSR_KERNEL_PANELS), (best, t)
        assert t[4] > 0 and t[2] > 0          # always candidates
        if dA.NZ / dA.M < 6:
            assert t[0] > 0, t                # thread_row was measured ...
            # ... and lost by > 2x (3.5-6x measured, with four loads in flight
            # per lane; 9-15x with one)
            assert ms < 0.5 * t[0], (ms, t)
        if kind == S.SYNTH_HUB:
            assert t[3] > 0, t                # block_row measured, and lost
            assert ms < 0.5 * t[3], (ms, t)
            # the hub row (131 072 entries) is spread over workgroups: the
            # stream kernel runs within 3x of the blocked pick, not 8x off
            assert t[4] < 3.0 * ms + 0.02, (ms, t)
            dH = dA.to_hll(True)
            bh, mh = dH.autotune(d_x.ptr, d_y.ptr)
            assert bh == S.HLL_KERNEL_PANELS     # the format pads 2.3x
            assert "long row(s) beside" in dH.panels_describe()
            assert mh < 2.0 * ms + 0.02, (mh, ms)   # was 4x the CSR time
            # the blocked copy stores no padding: priced on the true entries,
            # or a padded format would read > 100 % of the roofline
            assert dH.kernel_bytes(bh) < dH.algorithmic_bytes
            assert dH.kernel_bytes(bh) / (mh * 1e6) / 8000.0 < 1.0
        log = dA.tune_log()
        assert "direct kernels" in log and "total" in log
    finally:
        if dH is not None:
            dH.release()
        dA.release()
        d_x.free()
        d_y.free()
