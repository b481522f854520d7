"""Seeded random shapes through the 2-D blocked path (all schedules, random
tile heights and panel widths) and the direct kernels, against the oracle."""
import numpy as np
import pytest

import _oracle as O
import spmv_scpa_amd as S

pytestmark = pytest.mark.gpu

TIGHT = 1e-12


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = int(rng.choice([S.SYNTH_BANDED, S.SYNTH_RANDOM, S.SYNTH_RAGGED,
                               S.SYNTH_KKT, S.SYNTH_POWERLAW, S.SYNTH_HUB]))
        M = int(rng.choice([1, 31, 33, 257, 5_000, 20_011, 60_000]))
        N = int(rng.choice([1, 64, 4_097, 30_000, 150_000]))
        K = int(rng.integers(1, 41))
        if kind == S.SYNTH_BANDED:
            K = min(K, N)
        if kind == S.SYNTH_RAGGED and K < 4:
            K = 4
        if kind == S.SYNTH_KKT and K < 4:
            K = 4
        W = int(rng.choice([16, 1_000, 70_000, 1 << 30]))
        pc = int(rng.choice([0, 64, 1_000, 4_096]))
        tile = int(rng.choice([32, 96, 1_024, 8_192, 16_384, 20_448]))
        out.append((i, kind, M, N, K, W, pc, tile))
    return out


@pytest.mark.parametrize("case", _cases(int(__import__("os").environ.get("SPMV_FUZZ_CASES", "30")), int(__import__("os").environ.get("SPMV_FUZZ_SEED", "2024"))), ids=lambda c: "c%d" % c[0])
def test_blocked_path_random_shapes(case, monkeypatch):
    _, kind, M, N, K, W, pc, tile = case
    IRP, JA, AS = O.synth_csr(kind, M, N, K, W, 42)
    x = O.synth_x(7, 0, N)
    y_ref = O.csr_spmv(IRP, JA, AS, x)
    scale = O.csr_abs_spmv(IRP, JA, AS, x)
    A = S.csr_from_arrays("fuzz", M, N, IRP, JA, AS)
    d_x, d_y = S.DevBuffer.from_numpy(x), S.DevBuffer(max(M, 1) * 8)
    dA = S.CsrDevice.upload(A)
    dH = dA.to_hll(True)

    def check(tag):
        S.stream_sync()
        y = d_y.to_numpy(np.float64, M)
        err = np.max(np.abs(y - y_ref) / np.maximum(scale, 1e-300))
        assert err <= TIGHT, (case, tag, err)

    try:
        for k in (2, 4):
            S._lib.spmv_dev_memset(d_y.ptr, 0xFF, M * 8, None)
            dA.launch(k, d_x.ptr, d_y.ptr)
            check(("csr", k))
        # the workgroup orders of the direct kernels (variant bit 0: hardware
        # order, bit 1: XCD-contiguous ranges, bit 5 (CSR) / bit 2 (HLL):
        # grouped runs; stream kernel: bit 6 hardware order, bit 4 narrow loads)
        # bit 9: the sub-wave kernel loads IRP even when all rows have one
        # length (banded / fixed-degree cases take the computed-IRP path else)
        for k, variant in ((2, 1), (2, 2), (2, 32), (2, 512), (4, 32), (4, 64),
                           (4, 16 | 32)):
            S._lib.spmv_dev_memset(d_y.ptr, 0xFF, M * 8, None)
            dA.launch(k, d_x.ptr, d_y.ptr, variant=variant)
            check(("csr", k, "variant", variant))
        for k in (1, 2):
            for variant in (0, 1, 2, 4):
                S._lib.spmv_dev_memset(d_y.ptr, 0xFF, M * 8, None)
                dH.launch(k, d_x.ptr, d_y.ptr, variant=variant)
                check(("hll", k, "variant", variant))
        for sched in ("chain", "steps", "sweep"):
            S.set_panel_schedule(sched)
            if sched != "sweep":
                monkeypatch.setenv("SPMV_TILE_ROWS", str(tile))
                # odd cases: ascending bucket order instead of the residue
                # order banded shapes get by default
                monkeypatch.setenv("SPMV_BUCKET_ORDER", str(case[0] & 1))
            for m, blocked in ((dA, S.CSR_KERNEL_PANELS),
                               (dH, S.HLL_KERNEL_PANELS)):
                m.build_panels(pc)
                for waves in (0, 4):
                    S._lib.spmv_dev_memset(d_y.ptr, 0xFF, M * 8, None)
                    m.launch(blocked, d_x.ptr, d_y.ptr, waves_per_block=waves)
                    check((sched, blocked, waves))
            monkeypatch.delenv("SPMV_TILE_ROWS", raising=False)
            monkeypatch.delenv("SPMV_BUCKET_ORDER", raising=False)
    finally:
        S.set_panel_schedule("sweep")
        dH.release()
        dA.release()
        S.csr_free(A)
