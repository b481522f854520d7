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
