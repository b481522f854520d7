"""Perf floors that bite: parity tests do not notice a
kernel that got slower (round 5 lost 4x on the long rows' side launches for a
day).  The floors are NOT numbers typed into this file: `profiles/
r05_floors.json` holds, per kernel + workload, the MINIMUM of the fractions of
the 8 TB/s roofline measured on exclusive MI355X boxes (rounds 5 and 6, each
with its source), and a kernel must reach 0.80 x that -- box and clock spread
is +-10 %; a regression by a fifth fails.  Workloads at the sizes the records
were taken at (10M rows: the bench's own), event-timed medians of 20 launches
on the kernel's algorithmic bytes, as bench.py prices it.  Measured vs floor is
printed for every kernel -- also in the terminal summary of a quiet run
(conftest.pytest_terminal_summary) -- so the driver's log shows the margin.

    python tests/test_gpu_perf_floor.py --record out.json   # re-measure
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import spmv_scpa_amd as S  # noqa: E402

pytestmark = pytest.mark.gpu

PEAK = 8.0e12
FLUSH = 1 << 30
FLOORS = os.path.join(ROOT, "profiles", "r05_floors.json")
SLACK = 0.80


def _frac(m, kernel, x, y, flush=0):
    ms = float(np.median(m.time(kernel, x.ptr, y.ptr, 3, 20, flush)))
    return m.kernel_bytes(kernel) / (ms * 1e-3) / PEAK, ms


def measure_all():
    """-> {tag: (fraction of 8 TB/s, ms)} of the hot kernels"""
    M = N = 10_000_000
    x, y = S.DevBuffer(N * 8), S.DevBuffer(M * 8)
    S.dev_fill_synth(x.ptr, N, 7)
    seen = {}

    # banded 10M x 32: the literal north-star kernels
    dA = S.CsrDevice.generate(S.SYNTH_BANDED, M, N, 32, 0, 0, 42)
    dH = dA.to_hll(True)
    seen["banded10M hll_threads_col_major"] = _frac(dH, 1, x, y)
    seen["banded10M csr_stream"] = _frac(dA, 4, x, y)
    seen["banded10M csr_subwave_row"] = _frac(dA, 2, x, y)
    dH.release()
    dA.release()

    # random, W = 2^17: the blocked chain schedule, 8192-row tiles
    dA = S.CsrDevice.generate(S.SYNTH_RANDOM, M, N, 32, 1 << 17, 0, 42)
    dH = dA.to_hll(True)
    dA.release()
    dH.build_panels(0, "chain", 8192)
    seen["W=2^17 blocked chain 8192"] = _frac(dH, S.HLL_KERNEL_PANELS, x, y)
    dH.release()

    # random, W = 2^20: chain, the balanced tall tiles
    dA = S.CsrDevice.generate(S.SYNTH_RANDOM, M, N, 32, 1 << 20, 0, 42)
    dH = dA.to_hll(True)
    dA.release()
    dH.build_panels(0, "chain", 19552)
    seen["W=2^20 blocked chain 19552"] = _frac(dH, S.HLL_KERNEL_PANELS, x, y)
    dH.release()

    # random, columns anywhere (the headline): the sweep schedule as built by
    # default (deterministic since round 6) and the direct kernel it replaces
    dA = S.CsrDevice.generate(S.SYNTH_RANDOM, M, N, 32, 2 * N, 0, 42)
    dH = dA.to_hll(True)
    dA.release()
    dH.build_panels(0, "sweep")
    seen["W=N blocked sweep"] = _frac(dH, S.HLL_KERNEL_PANELS, x, y)
    seen["W=N hll_threads_col_major"] = _frac(dH, 1, x, y)
    dH.release()

    # config 2: 1M x 16 banded CSR, flushed
    dA = S.CsrDevice.generate(S.SYNTH_BANDED, 1_000_000, 1_000_000, 16, 0, 0, 42)
    seen["config2 csr_stream flushed"] = _frac(dA, 4, x, y, FLUSH)
    seen["config2 csr_subwave_row flushed"] = _frac(dA, 2, x, y, FLUSH)
    dA.release()
    x.free()
    y.free()
    return seen


def test_perf_floors_of_the_hot_kernels(request):
    rec = json.load(open(FLOORS))["floors"]
    seen = measure_all()
    assert set(seen) == set(rec), (sorted(seen), sorted(rec))
    bad = {}
    for tag, (frac, ms) in seen.items():
        floor = SLACK * rec[tag]["measured_min"]
        line = ("%-36s %.4f ms  %.3f of 8 TB/s  floor %.3f (= %.2f x measured "
                "min %.3f)  margin %+.0f %%"
                % (tag, ms, frac, floor, SLACK, rec[tag]["measured_min"],
                   100.0 * (frac / floor - 1.0)))
        print(line)
        # also after the run's summary: visible in a quiet, captured run
        getattr(request.config, "_summary_lines", []).append(line)
        if frac < floor:
            bad[tag] = (round(frac, 4), round(floor, 4))
    assert not bad, bad


if __name__ == "__main__":
    out = {t: {"frac": round(f, 4), "ms": round(ms, 5)}
           for t, (f, ms) in measure_all().items()}
    if len(sys.argv) > 2 and sys.argv[1] == "--record":
        json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))
