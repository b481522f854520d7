"""Perf floors: parity tests do not notice a kernel that got 4x slower (round
5 lost 4x on the long rows' side launches for a day).  Each floor is ~55 % of
what the kernel measures on an exclusive MI355X -- loose enough for clock and
box spread (+-10 %), tight enough for a regression by half.  Event-timed medians
of 20 launches, fraction of the 8 TB/s roofline on the kernel's algorithmic
bytes, as bench.py prices it."""
import numpy as np
import pytest

import spmv_scpa_amd as S

pytestmark = pytest.mark.gpu

PEAK = 8.0e12
FLUSH = 1 << 30


def _frac(m, kernel, x, y, flush=0):
    ms = float(np.median(m.time(kernel, x.ptr, y.ptr, 3, 20, flush)))
    return m.kernel_bytes(kernel) / (ms * 1e-3) / PEAK, ms


def test_perf_floors_of_the_hot_kernels():
    M = N = 4_000_000
    x, y = S.DevBuffer(N * 8), S.DevBuffer(M * 8)
    S.dev_fill_synth(x.ptr, N, 7)
    seen = {}

    # banded: the literal north-star kernels (measured 0.76-0.85 / 0.68-0.78)
    dA = S.CsrDevice.generate(S.SYNTH_BANDED, M, N, 32, 0, 0, 42)
    dH = dA.to_hll(True)
    seen["banded hll_threads_col_major"] = _frac(dH, 1, x, y) + (0.42,)
    seen["banded csr_stream"] = _frac(dA, 4, x, y) + (0.38,)
    seen["banded csr_subwave_row"] = _frac(dA, 2, x, y) + (0.33,)
    dH.release()
    dA.release()

    # random, W = 2^17: the blocked chain schedule (0.80-0.84 at 10M rows)
    dA = S.CsrDevice.generate(S.SYNTH_RANDOM, M, N, 32, 1 << 17, 0, 42)
    dH = dA.to_hll(True)
    dA.release()
    dH.build_panels(0, "chain", 8192)
    seen["W=2^17 blocked chain"] = _frac(dH, S.HLL_KERNEL_PANELS, x, y) + (0.40,)
    dH.release()

    # random, columns anywhere: the sweep schedule (0.42 at 5M rows, 0.345 at
    # 10M) and the direct kernel it replaces (fabric-bound, 0.085)
    dA = S.CsrDevice.generate(S.SYNTH_RANDOM, M, N, 32, 2 * N, 0, 42)
    dH = dA.to_hll(True)
    dA.release()
    dH.build_panels(0, "sweep")
    seen["W=N blocked sweep"] = _frac(dH, S.HLL_KERNEL_PANELS, x, y) + (0.24,)
    seen["W=N hll_threads_col_major"] = _frac(dH, 1, x, y) + (0.05,)
    dH.release()

    # config 2: 1M x 16 banded CSR, flushed (0.54-0.60 / 0.60-0.71)
    dA = S.CsrDevice.generate(S.SYNTH_BANDED, 1_000_000, 1_000_000, 16, 0, 0, 42)
    seen["config2 csr_stream flushed"] = _frac(dA, 4, x, y, FLUSH) + (0.33,)
    seen["config2 csr_subwave_row flushed"] = _frac(dA, 2, x, y, FLUSH) + (0.30,)
    dA.release()

    for tag, (frac, ms, floor) in seen.items():
        print("%-36s %.4f ms  %.3f of 8 TB/s  (floor %.2f)" % (tag, ms, frac,
                                                               floor))
    bad = {t: v for t, v in seen.items() if v[0] < v[2]}
    assert not bad, bad
    x.free()
    y.free()
