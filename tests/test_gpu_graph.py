"""Launches are stream-ordered and capturable: a caller that iterates
y = A x (a solver) can record the launches of a handle into a hipGraph once
and replay it -- torch.cuda.CUDAGraph here, hipStreamBeginCapture underneath.
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
    import torch
    M = N = 300_000
    dev = torch.device("cuda", 0)
    dA = S.CsrDevice.generate(kind, M, N, K, W, 0, 42)
    dHc, dHr = dA.to_hll(True), dA.to_hll(False)
    x = torch.empty(N, dtype=torch.float64, device=dev)
    y = torch.zeros(M, dtype=torch.float64, device=dev)
    S.dev_fill_synth(x.data_ptr(), N, 7, 0,
                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    cases = [(dA, k, "csr%d" % k) for k in range(S.NUM_CSR_KERNELS)]
    cases += [((dHc if S.HLL_KERNEL_COL_MAJOR[k] else dHr), k, "hll%d" % k)
              for k in range(S.NUM_HLL_KERNELS)]
    for sched in ("chain", "steps", "sweep"):
        cases.append((dA, S.CSR_KERNEL_PANELS, "csr blocked " + sched))
        cases.append((dHc, S.HLL_KERNEL_PANELS, "hll blocked " + sched))
    side = torch.cuda.Stream()
    for m, kernel, tag in cases:
        blocked = "blocked" in tag
        if blocked:
            m.build_panels(0, tag.split()[-1], deterministic=True)
        # eager reference (also sets the one-off function attributes)
        m.launch(kernel, x.data_ptr(), y.data_ptr(),
                 stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want = y.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(2):  # two SpMVs per replay, back to back
                m.launch(kernel, x.data_ptr(), y.data_ptr(),
                         stream=torch.cuda.current_stream().cuda_stream)
        for rep in range(4):
            y.fill_(float("nan"))
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(y, want), (tag, rep)
        # the graph reads x where it lives: new values, same graph
        x.mul_(2.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, want * 2.0), tag  # exact: a power of two
        x.mul_(0.5)
        del g
    # and against the oracle's rows, once
    rows = np.random.default_rng(3).integers(0, M, 64)
    got = want.cpu().numpy()
    for r in rows:
        w, sc = O.synth_row_dot(kind, M, N, K, W, 0, 42, 7, int(r))
        assert abs(got[r] - w) <= 1e-12 * sc, r
    for m in (dHc, dHr, dA):
        m.release()
