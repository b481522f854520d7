"""BASELINE config 4 (SuiteSparse nlpkkt160: CSR, 8.3 M rows, ~2.3e8 nnz,
irregular rows) at REAL SIZE through the real path:

    .mtx text -> io_load_csr_cached -> sparse_csr -> spmv_csr_upload ->
    CSR kernels 1 (wavefront per row), 2 (sub-wavefront), 4 (stream) and the
    autotuned pick -> the WHOLE y against the oracle's serial CSR loop.

The reference's workflow is .mtx-driven (src/main.c:78; matrix list
scripts/download-matrices.py:7-38).  nlpkkt160.mtx cannot be fetched here, so
tools/gen_kkt_mtx.c writes a symmetric real coordinate file of its shape
(M = 8 345 600, KKT blocks [H A'; A 0] of a 160^3 grid, 1.18e8 stored entries
-> 2.31e8 after mirroring) into the box's temp dir; if
$SPMV_MTX_DIR/nlpkkt160.mtx exists that file runs instead.  Loader parity
with the reference's semantics (file order, mirroring; src/csr.c:31-171) is
pinned bit for bit on smaller files of the same generator in
tests/test_kkt_mtx.py; here rows are re-derived from the matrix definition,
independent of any parser.

SPMV_KKT_N overrides the grid edge (default 160) for quick runs.
"""
import os
import time

import numpy as np
import pytest

import _kkt as K
import _oracle as O
import spmv_scpa_amd as S

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6   # BASELINE north_star: relative fp64 vs serial CSR
TIGHT = 1e-12    # of the row scale sum |a_ij x_j|: only the order may differ


@pytest.fixture(scope="module")
def kkt(tmp_path_factory):
    real = os.path.join(os.environ.get("SPMV_MTX_DIR", ""), "nlpkkt160.mtx")
    if os.environ.get("SPMV_MTX_DIR") and os.path.exists(real):
        return dict(path=real, n=None)
    n = int(os.environ.get("SPMV_KKT_N", "160"))
    d = tmp_path_factory.mktemp("config4")
    t0 = time.time()
    p = K.write_mtx(n, str(d / ("kkt%d.mtx" % n)))
    print("\n[config4] wrote %s (%.2f GB) in %.1f s"
          % (p, os.path.getsize(p) / 1e9, time.time() - t0))
    return dict(path=p, n=n)


def test_config4_loader_to_kernels_whole_y(kkt):
    p, n = kkt["path"], kkt["n"]
    t0 = time.time()
    A = S.io_load_csr_cached(p)
    t_text = time.time() - t0
    M, N, NZ = A.contents.M, A.contents.N, A.contents.NZ
    S.csr_free(A)
    t0 = time.time()
    A = S.io_load_csr_cached(p)  # now from the validated .bin sidecar
    t_bin = time.time() - t0
    assert (A.contents.M, A.contents.N, A.contents.NZ) == (M, N, NZ)
    print("[config4] M=%d NZ=%d  text load %.2f s (incl. sidecar write), "
          ".bin load %.2f s" % (M, NZ, t_text, t_bin))
    IRP, JA, AS = S.csr_arrays(A)
    if n is not None:
        eM, stored, nnz = K.expected_counts(n)
        assert (M, N, NZ) == (eM, eM, nnz)
        if n == 160:
            assert M == 8_345_600 and 2.2e8 < NZ < 2.4e8
    lens = np.diff(IRP)
    assert lens.min() >= 0 and lens.max() > lens.min()  # irregular rows

    # x as the reference fills it for .mtx runs (src/vector.c: rand()/RAND_MAX)
    x = S.vec_random(N)
    y_ref = O.csr_spmv(IRP, JA, AS, x)          # serial CSR: the oracle
    scale = O.csr_abs_spmv(IRP, JA, AS, x)
    if n is not None:  # rows from the definition, no parser involved
        rng = np.random.default_rng(4)
        G, B, n1, _ = K.dims(n)
        rows = np.unique(np.concatenate([[0, G - 1, G, n1 - 1, n1, M - 1],
                                         rng.integers(0, M, 2000)]))
        for i in rows:
            d, s = K.row_dot(n, int(i), x)
            assert abs(d - y_ref[i]) <= TIGHT * max(s, 1e-300), i

    dA = S.CsrDevice.upload(A)
    d_x, d_y = S.DevBuffer.from_numpy(x), S.DevBuffer(M * 8)
    best, best_ms = dA.autotune(d_x.ptr, d_y.ptr)
    print("[config4] autotuned pick: csr_%s  %.3f ms = %.1f%% of the HBM "
          "roofline" % (S.CSR_KERNEL_LABELS[best], best_ms,
                        100 * dA.algorithmic_bytes / (best_ms * 1e6) / 8000))
    den = np.maximum(np.maximum(np.abs(y_ref), 1e-3 * scale), 1e-300)
    for tag, k in (("wave_row", 1), ("subwave_row", 2), ("stream", 4),
                   ("autotuned", best)):
        S._lib.spmv_dev_memset(d_y.ptr, 0xFF, M * 8, None)
        dA.launch(k, d_x.ptr, d_y.ptr)
        S.stream_sync()
        y = d_y.to_numpy(np.float64, M)
        assert np.all(np.isfinite(y)), tag
        assert np.max(np.abs(y - y_ref) / den) <= REL_TOL, tag
        assert np.linalg.norm(y - y_ref) <= REL_TOL * np.linalg.norm(y_ref), tag
        assert np.max(np.abs(y - y_ref) / np.maximum(scale, 1e-300)) <= TIGHT, tag
        assert S.validation_vec_result(y_ref, y) == 0, tag  # reference's -d
        ms = float(np.median(dA.time(k, d_x.ptr, d_y.ptr, warmup=2, iters=10)))
        print("[config4] csr_%-12s %.3f ms  %.1f GFLOP/s  %.1f%% of roofline"
              % (tag, ms, 2.0 * NZ / (ms * 1e6),
                 100 * dA.algorithmic_bytes / (ms * 1e6) / 8000))
    dA.release()
    S.csr_free(A)
