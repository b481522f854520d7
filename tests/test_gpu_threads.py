"""INTEGRATION.md, "Handles are re-entrant per handle, launches are
stream-ordered per handle": two host threads, each with its own handle and
its own stream, launch at the same time.  The matrices are the ones with state
that a launch leaves behind for the next (arrival counters of the long rows'
segments, of the wide hack blocks, of the rows beside the blocked copy): a
counter shared between handles, or a launch that strays onto the default
stream, shows up as a wrong or a changing y."""
import threading

import numpy as np
import pytest

import _oracle as O
import spmv_scpa_amd as S

pytestmark = pytest.mark.gpu


def test_two_host_threads_two_handles_two_streams():
    specs = [(S.SYNTH_HUB, 200_000, 6, 4096), (S.SYNTH_POWERLAW, 300_000, 8, 1 << 30)]
    work = []
    for kind, M, K, W in specs:
        dA = S.CsrDevice.generate(kind, M, M, K, W, 0, 42)
        dH = dA.to_hll(True)
        dA.build_panels(0, "chain")
        d_x, d_y = S.DevBuffer(M * 8), S.DevBuffer(M * 8)
        S.dev_fill_synth(d_x.ptr, M, 7)
        rows = np.unique(np.concatenate([[0, M // 3, M - 1],
                                         np.random.default_rng(1).integers(0, M, 200)]))
        want = np.array([O.synth_row_dot(kind, M, M, K, W, 0, 42, 7, int(g))
                         for g in rows])
        ref = {}
        for tag, h, k in (("csr stream", dA, 4), ("csr sub-wave", dA, 2),
                          ("hll col", dH, 1), ("blocked", dA, S.CSR_KERNEL_PANELS)):
            h.launch(k, d_x.ptr, d_y.ptr)
            S.stream_sync()
            y = d_y.to_numpy(np.float64, M)
            assert np.max(np.abs(y[rows] - want[:, 0]) / want[:, 1]) <= 1e-12, tag
            ref[tag] = (h, k, y)
        work.append((M, d_x, d_y, ref, S.Stream(), dA, dH))
    errors = []
    start = threading.Barrier(len(work))

    def worker(M, d_x, d_y, ref, stream, *_):
        st = stream.ptr
        try:
            start.wait()
            for it in range(60):
                for tag, (h, k, y_ref) in ref.items():
                    h.launch(k, d_x.ptr, d_y.ptr, stream=st)
                    h.launch(k, d_x.ptr, d_y.ptr, stream=st)  # counters re-armed
                    S.stream_sync(st)
                    y = d_y.to_numpy(np.float64, M)
                    if tag == "blocked":   # LDS atomics: equal to rounding
                        bad = np.max(np.abs(y - y_ref)) > 1e-11 * (1 + np.max(np.abs(y_ref)))
                    else:                  # fixed summation orders: equal bits
                        bad = not np.array_equal(y, y_ref)
                    if bad:
                        errors.append((tag, it))
                        return
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=w) for w in work]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for M, d_x, d_y, ref, stream, dA, dH in work:
        dH.release()
        dA.release()
        d_x.free()
        d_y.free()


def test_two_host_threads_upload_build_and_tune_at_the_same_time():
    """... "may upload, build, tune and launch DIFFERENT handles
    concurrently": generation, CSR -> HLL, the selector (its block pool and
    phase log are per thread, its sorts share the default stream) and the
    launch of its pick, in two threads at once; y against the oracle's rows.
    What the selectors pick under each other's load is not the point."""
    specs = [(S.SYNTH_HUB, 400_000, 6, 4096), (S.SYNTH_RANDOM, 600_000, 32, 1 << 30)]
    errors = []
    start = threading.Barrier(len(specs))

    def worker(kind, M, K, W):
        try:
            rows = np.unique(np.concatenate(
                [[0, M // 3, M - 1], np.random.default_rng(2).integers(0, M, 300)]))
            want = np.array([O.synth_row_dot(kind, M, M, K, W, 0, 42, 7, int(g))
                             for g in rows])
            start.wait()
            for rep in range(3):
                dA = S.CsrDevice.generate(kind, M, M, K, W, 0, 42)
                dH = dA.to_hll(True)
                d_x, d_y = S.DevBuffer(M * 8), S.DevBuffer(M * 8)
                S.dev_fill_synth(d_x.ptr, M, 7)
                for h in (dA, dH):
                    best, ms = h.autotune(d_x.ptr, d_y.ptr)
                    assert ms > 0 and "total" in h.tune_log()
                    S._lib.spmv_dev_memset(d_y.ptr, 0xFF, M * 8, None)
                    h.launch(best, d_x.ptr, d_y.ptr)
                    S.stream_sync()
                    y = d_y.to_numpy(np.float64, M)
                    err = np.max(np.abs(y[rows] - want[:, 0]) / want[:, 1])
                    assert err <= 1e-12, (kind, rep, best, err)
                dH.release()
                dA.release()
                d_x.free()
                d_y.free()
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=s) for s in specs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
