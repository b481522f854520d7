"""Launches are stream-ordered and capturable: a caller that iterates
y = A x (a solver) can record the launches of a handle into a hipGraph once
and replay it -- through the library's own C-ABI (spmv_stream_create,
spmv_graph_begin_capture / _end_capture / _launch, spmv_engine.h), no torch
in the process: the test runs on the ROCm runtime the library was built for.
Every kernel id of both formats, with everything a launch may consist of: the
main kernel, the segmented side launch of long rows / wide hack blocks (whose
last-arriver counters carry a launch number that a replay REPEATS:
epoch_arrive / epoch_rearm, hip_common.h), the memset + persistent kernel of
the sweep schedule, the multi-launch steps schedule."""
import numpy as np
import pytest

import _oracle as O
import spmv_scpa_amd as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,K,W", [(S.SYNTH_HUB, 6, 4096),
                                      (S.SYNTH_RANDOM, 32, 1 << 30)],
                         ids=["hub", "random"])
def test_every_launch_replays_from_a_captured_graph(kind, K, W):
    M = N = 300_000
    dA = S.CsrDevice.generate(kind, M, N, K, W, 0, 42)
    dHc, dHr = dA.to_hll(True), dA.to_hll(False)
    x, y = S.DevBuffer(N * 8), S.DevBuffer(M * 8)
    S.dev_fill_synth(x.ptr, N, 7)
    S.stream_sync()
    cases = [(dA, k, "csr%d" % k) for k in range(S.NUM_CSR_KERNELS)]
    cases += [((dHc if S.HLL_KERNEL_COL_MAJOR[k] else dHr), k, "hll%d" % k)
              for k in range(S.NUM_HLL_KERNELS)]
    for sched in ("chain", "steps", "sweep"):
        cases.append((dA, S.CSR_KERNEL_PANELS, "csr blocked " + sched))
        cases.append((dHc, S.HLL_KERNEL_PANELS, "hll blocked " + sched))
    side = S.Stream()

    def eager(m, kernel):
        m.launch(kernel, x.ptr, y.ptr, stream=side.ptr)
        side.sync()
        return y.to_numpy(np.float64, M)

    want = None
    for m, kernel, tag in cases:
        if "blocked" in tag:
            m.build_panels(0, tag.split()[-1], deterministic=True)
        # eager reference (also sets the one-off function attributes)
        want = eager(m, kernel)
        with side.capture() as g:
            for _ in range(2):  # two SpMVs per replay, back to back
                m.launch(kernel, x.ptr, y.ptr, stream=side.ptr)
        for rep in range(4):
            S._check(S._lib.spmv_dev_memset(y.ptr, 0xFF, M * 8, side.ptr),
                     "spmv_dev_memset")  # NaNs
            g.launch(side.ptr)
            side.sync()
            got = y.to_numpy(np.float64, M)
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (
                tag, rep)
        # the graph reads x where it lives: new values, same graph
        S.dev_fill_synth(x.ptr, N, 8, 0, side.ptr)
        want2 = eager(m, kernel)
        assert not np.array_equal(want2, want)
        S._check(S._lib.spmv_dev_memset(y.ptr, 0xFF, M * 8, side.ptr),
                 "spmv_dev_memset")
        g.launch(side.ptr)
        side.sync()
        assert np.array_equal(y.to_numpy(np.float64, M).view(np.uint64),
                              want2.view(np.uint64)), tag
        S.dev_fill_synth(x.ptr, N, 7, 0, side.ptr)
        side.sync()
        g.destroy()
    # and against the oracle's rows, once
    rows = np.random.default_rng(3).integers(0, M, 64)
    for r in rows:
        w, sc = O.synth_row_dot(kind, M, N, K, W, 0, 42, 7, int(r))
        assert abs(want[r] - w) <= 1e-12 * sc, r
    for m in (dHc, dHr, dA):
        m.release()
    x.free()
    y.free()


def test_the_default_stream_cannot_be_captured():
    assert S._lib.spmv_graph_begin_capture(None) == -22  # -EINVAL
