"""Drop-in proof on the GPU: the reference's UNMODIFIED driver and host layer
(main.c, csr.c, hll.c, logger.c, ... compiled from /root/reference/src by
oracle/build_ref.sh into oracle/_ref/ref_dropin) linked against
libspmv_scpa_amd.so, which supplies the 11 plugin symbols the reference
declares in cuda_csr.h / cuda_hll.h.  The reference's own -d validation
(serial CSR vs every GPU variant, main.c:282-293, 337-348) must pass and its
cuda.csv must hold the full 27-row grid."""
import os
import subprocess

import pytest

import _golden as G
import spmv_scpa_amd as S

pytestmark = pytest.mark.gpu

DROPIN = os.path.join(S.ROOT, "oracle", "_ref", "ref_dropin")


@pytest.mark.parametrize("name", ["gen", "cage4_like", "ragged100", "sym70",
                                  "tail40", "hub96"])
def test_reference_driver_runs_on_mi355x_kernels(name, tmp_path):
    if not os.path.exists(DROPIN):
        pytest.skip("oracle/_ref/ref_dropin not built (needs /root/reference "
                    "at build time)")
    out = str(tmp_path)
    # the reference asserts num_threads <= omp_get_max_threads() for its
    # 2..40 thread ladder (hll.c:184)
    env = dict(os.environ, OMP_NUM_THREADS="40")
    r = subprocess.run([DROPIN, "-m", G.mtx_path(name), "-o", out, "-d"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    rows = open(os.path.join(out, "cuda.csv")).read().splitlines()
    assert rows[0] == ("matrix,format,kernel,warps_per_block,rows,cols,nnz,"
                       "num_blocks,duration_ms,gflops")
    assert len(rows) == 1 + 5 * 3 + 4 * 3  # reference main.c:258-354
    for line in rows[1:]:
        f = line.split(",")
        assert f[0] == name and f[1] in ("CSR", "HLL")
        assert float(f[-2]) > 0.0  # a real kernel time, not an error code


PRODUCT_DRIVER = os.path.join(S.ROOT, "spmv_scpa_amd", "bin", "spmv_scpa_amd")


def test_product_driver_full_grid_on_gpu(tmp_path):
    """This repo's own driver (same flags and CSV files as the reference's):
    CPU grid + 27-row one-shot GPU grid with -d validation + resident timing
    with roofline.csv + measured kernel choice."""
    out = str(tmp_path)
    env = dict(os.environ, OMP_NUM_THREADS="8")
    for args in (["-m", G.mtx_path("ragged100")],
                 ["-s", "random", "--rows", "300000", "--nnz-row", "32",
                  "--window", "100000"]):
        r = subprocess.run([PRODUCT_DRIVER] + args + ["-o", out, "-d",
                                                      "--iters", "5"],
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "autotune" in r.stdout
    gpu = open(os.path.join(out, "cuda.csv")).read().splitlines()
    assert len(gpu) == 1 + 2 * 27
    roof = open(os.path.join(out, "roofline.csv")).read().splitlines()
    assert roof[0].startswith("matrix,format,kernel,waves_per_block,gpus,")
    assert len(roof) >= 1 + 2 * 9
    for line in roof[1:]:
        f = line.split(",")
        assert float(f[-4]) > 0 and 0 < float(f[-1]) < 1.0


@pytest.mark.parametrize("family,rows,k,window", [
    ("hub", 200_000, 6, 512),         # dc1 class: a row of 131 072 entries
    ("powerlaw", 300_000, 3, 0),      # webbase / amazon / roadNet class
])
def test_product_driver_on_the_irregular_classes(family, rows, k, window,
                                                 tmp_path):
    """the 27-row one-shot grid (every seam kernel called by name, -d
    validation against serial CSR like the reference's driver) on the matrix
    classes where a kernel used to walk a hub row with one lane: every call
    must validate, and none may take the milliseconds it used to"""
    out = str(tmp_path)
    env = dict(os.environ, OMP_NUM_THREADS="8")
    r = subprocess.run([PRODUCT_DRIVER, "-s", family, "--rows", str(rows),
                        "--nnz-row", str(k), "--window", str(window),
                        "-o", out, "-d", "--iters", "3", "--no-cpu"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    gpu = open(os.path.join(out, "cuda.csv")).read().splitlines()
    assert len(gpu) == 1 + 27  # every one-shot call validated (-d) and logged
    # the warm, resident timing of every kernel id (roofline.csv: medians of
    # --iters launches; the one-shot numbers above are single cold launches on
    # freshly uploaded memory, like the reference's): the hub row alone cost
    # 6-16 ms per launch before the long-row / wide-block launches
    roof = open(os.path.join(out, "roofline.csv")).read().splitlines()
    times = {(f[1], int(f[2])): float(f[10])
             for f in (line.split(",") for line in roof[1:])}
    assert len(times) >= 9, times
    slow = {k: v for k, v in times.items() if v > 2.0}
    assert not slow, slow


DROPIN_CACHED = os.path.join(S.ROOT, "oracle", "_ref", "ref_dropin_cached")


@pytest.mark.parametrize("name", ["ragged100", "hub96", "tail40"])
def test_reference_driver_with_the_seam_cache_on(name, tmp_path):
    """the reference's unmodified driver + oracle/seam_cache_on.c (the one
    line `spmv_seam_cache(2)` as a constructor): its own -d validation passes
    on all 27 calls and the CSVs name the same grid as the uncached run"""
    if not os.path.exists(DROPIN_CACHED):
        pytest.skip("oracle/_ref/ref_dropin_cached not built")
    env = dict(os.environ, OMP_NUM_THREADS="40")
    grids = []
    for exe, sub in ((DROPIN, "plain"), (DROPIN_CACHED, "cached")):
        out = tmp_path / sub
        out.mkdir()
        r = subprocess.run([exe, "-m", G.mtx_path(name), "-o", str(out), "-d"],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        rows = open(out / "cuda.csv").read().splitlines()
        assert len(rows) == 1 + 27
        grids.append([line.split(",")[:8] for line in rows])
        for line in rows[1:]:
            assert float(line.split(",")[-2]) > 0.0
    assert grids[0] == grids[1]


def test_seam_cache_keeps_the_last_upload_and_notices_a_change():
    """spmv_seam_cache (hip_csr.h): hits on the same matrix, bit-equal y,
    a miss when the matrix changes (other struct, other content at the head),
    x re-uploaded when it changes; level 0 releases every handle; and a call
    on a 1M-row matrix costs about the kernel + the y download, not the
    6 ms of upload it costs uncached."""
    import time

    import numpy as np

    import _oracle as O
    M = N = 1_000_000
    A = S.csr_generate(S.SYNTH_BANDED, M, N, 16, 0, 0, 42)
    IRP, JA, AS = S.csr_arrays(A)
    x = O.synth_x(7, 0, N)
    base = S._lib.spmv_live_handles()
    y0, k0 = S.csr_spmv_hip(A, x, kernel=4)         # uncached
    t0 = time.perf_counter()
    S.csr_spmv_hip(A, x, kernel=4)                  # uncached, allocations warm
    wall_uncached = (time.perf_counter() - t0) * 1e3
    assert S._lib.spmv_live_handles() == base
    S.seam_cache(2)
    try:
        held0, h0, m0 = S.seam_cache_stats()
        y1, _ = S.csr_spmv_hip(A, x, kernel=4)      # miss: uploads, keeps
        # the timed calls go straight to the C entry point with ONE caller
        # buffer for y: a fresh 8 MB numpy array per call costs milliseconds
        # of page faults once the process has churned through gigabytes (the
        # config-4 test earlier in the suite), which is not the seam's time
        import ctypes as C
        y2 = np.full(M, -1.0)  # pages touched before the clock starts
        xp = x.ctypes.data_as(C.POINTER(C.c_double))
        yp = y2.ctypes.data_as(C.POINTER(C.c_double))
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            kms = S._lib.csr_spmv_hip_stream(A, xp, yp, None)
        wall = (time.perf_counter() - t0) * 1e3 / reps
        assert kms > 0
        held, h, m = S.seam_cache_stats()
        assert (held, h - h0, m - m0) == (1, reps, 1)
        assert np.array_equal(y0, y1) and np.array_equal(y0, y2)
        assert S._lib.spmv_live_handles() == base + 1
        # other kernels of the same matrix hit too
        for k in (0, 1, 2, 3):
            yk, _ = S.csr_spmv_hip(A, x, kernel=k)
            assert np.allclose(yk, y0, rtol=1e-12, atol=1e-12)
        assert S.seam_cache_stats()[1] - h0 == reps + 4
        # x changed in another buffer -> uploaded; same buffer, new content
        # at the head -> the fingerprint notices
        x2 = x * 2.0
        y3, _ = S.csr_spmv_hip(A, x2, kernel=4)
        assert np.allclose(y3, 2.0 * y0, rtol=1e-13, atol=0)
        x2[:8] = 0.0
        y4, _ = S.csr_spmv_hip(A, x2, kernel=4)
        ref = O.csr_spmv(np.array(IRP), np.array(JA), np.array(AS), x2)
        assert np.allclose(y4, ref, rtol=1e-12, atol=1e-12)
        # the matrix changed in place (first value): re-uploaded, not stale
        AS[0] *= 3.0
        m_before = S.seam_cache_stats()[2]
        y5, _ = S.csr_spmv_hip(A, x, kernel=4)
        assert S.seam_cache_stats()[2] == m_before + 1
        ref = O.csr_spmv(np.array(IRP), np.array(JA), np.array(AS), x)
        assert np.allclose(y5, ref, rtol=1e-12, atol=1e-12)
        # HLL slots: row-major and col-major copies are held side by side
        Hr, Hc = S.csr_to_hll(A, False), S.csr_to_hll(A, True)
        for _ in range(2):
            yr, _ = S.hll_spmv_hip(Hr, x, kernel=0)
            yc, _ = S.hll_spmv_hip(Hc, x, kernel=1)
            assert np.allclose(yr, ref, rtol=1e-12, atol=1e-12)
            assert np.allclose(yc, ref, rtol=1e-12, atol=1e-12)
        assert S.seam_cache_stats()[0] == 3
        assert S._lib.spmv_live_handles() == base + 3
        S.hll_free(Hr)
        S.hll_free(Hc)
    finally:
        S.seam_cache(0)
    assert S.seam_cache_stats()[0] == 0
    assert S._lib.spmv_live_handles() == base
    S.csr_free(A)
    # a resident call = x is there (level 2), memset y, one event-timed launch,
    # 8 MB of y back over PCIe: well under a millisecond of host time beyond
    # the download; uncached it was 6.3 ms (DESIGN, one-shot seam)
    print("seam cache: %.3f ms per call (uncached %.3f), kernel %.4f ms"
          % (wall, wall_uncached, kms))
    # 0.7 vs 6.4 ms on a quiet box; on a busy host both stretch
    assert wall < 2.5 or wall < 0.6 * wall_uncached, (wall, wall_uncached)


def test_seam_cache_level_3_cannot_be_stale_and_levels_1_2_say_what_they_miss():
    """VERDICT r05 next #6.  One entry in the MIDDLE of AS -- outside the 64
    head + 64 tail + 64 strided samples of the level-1/2 fingerprint -- is
    edited in place between two calls on the same struct:
      level 2  returns the PREVIOUS y (the documented contract of the sampled
               levels: the caller promised not to do that) -- until
               spmv_seam_cache_invalidate(A) is called;
      level 3  hashes every byte on every call: it re-uploads by itself.
    Same for x: an unsampled element edited in place."""
    import time

    import numpy as np

    import _oracle as O
    M = N = 1_000_000
    A = S.csr_generate(S.SYNTH_BANDED, M, N, 16, 0, 0, 42)
    IRP, JA, AS = S.csr_arrays(A)
    NZ = len(AS)
    x = O.synth_x(7, 0, N)
    step = NZ // 64
    spot = 37 * step + step // 2 + 3     # between two strided samples
    assert spot % step and 64 <= spot < NZ - 64
    row = int(np.searchsorted(np.array(IRP), spot, side="right") - 1)
    xs = 500_003                          # x: N // 64 = 15625; not a sample
    assert xs % (N // 64) and 64 <= xs < N - 64

    def ref():
        return O.csr_spmv(np.array(IRP), np.array(JA), np.array(AS), x)

    try:
        # ---- level 2: sampled fingerprints, the stale result and its cure
        S.seam_cache(2)
        y0, _ = S.csr_spmv_hip(A, x, kernel=4)
        assert np.allclose(y0, ref(), rtol=1e-12, atol=1e-12)
        AS[spot] += 1000.0
        y_stale, _ = S.csr_spmv_hip(A, x, kernel=4)
        assert np.array_equal(y_stale, y0)          # the contract's fine print
        assert abs(ref()[row] - y0[row]) > 1.0      # ... and it IS stale
        assert S.seam_cache_invalidate(A) == 1
        y_new, _ = S.csr_spmv_hip(A, x, kernel=4)
        assert np.allclose(y_new, ref(), rtol=1e-12, atol=1e-12)
        x[xs] += 7.0
        y_xstale, _ = S.csr_spmv_hip(A, x, kernel=4)
        assert np.array_equal(y_xstale, y_new)
        assert S.seam_cache_invalidate(x) == 1      # x only: the matrix stays
        m_before = S.seam_cache_stats()[2]
        y_x, _ = S.csr_spmv_hip(A, x, kernel=4)
        assert S.seam_cache_stats()[2] == m_before  # a hit, x re-uploaded
        assert np.allclose(y_x, ref(), rtol=1e-12, atol=1e-12)
        # ---- level 3: every byte, every call
        S.seam_cache(3)
        assert S.seam_cache_stats()[0] == 0  # keyed under the other function
        y3, _ = S.csr_spmv_hip(A, x, kernel=4)
        AS[spot] -= 500.0
        h0, m0 = S.seam_cache_stats()[1:]
        y3b, _ = S.csr_spmv_hip(A, x, kernel=4)
        assert S.seam_cache_stats()[2] == m0 + 1    # noticed: re-uploaded
        assert np.allclose(y3b, ref(), rtol=1e-12, atol=1e-12)
        assert abs(y3b[row] - y3[row]) > 1.0
        x[xs] -= 3.0
        y3c, _ = S.csr_spmv_hip(A, x, kernel=4)
        assert S.seam_cache_stats()[1] == h0 + 1    # matrix hit, x re-sent
        assert np.allclose(y3c, ref(), rtol=1e-12, atol=1e-12)
        # what level 3 costs per call on config-2's 212 MB (the C entry point,
        # one caller buffer for y)
        import ctypes as C
        y2 = np.full(M, -1.0)
        xp = x.ctypes.data_as(C.POINTER(C.c_double))
        yp = y2.ctypes.data_as(C.POINTER(C.c_double))
        t0 = time.perf_counter()
        for _ in range(5):
            S._lib.csr_spmv_hip_stream(A, xp, yp, None)
        ms3 = (time.perf_counter() - t0) * 1e3 / 5
        S.seam_cache(2)
        S._lib.csr_spmv_hip_stream(A, xp, yp, None)
        t0 = time.perf_counter()
        for _ in range(5):
            S._lib.csr_spmv_hip_stream(A, xp, yp, None)
        ms2 = (time.perf_counter() - t0) * 1e3 / 5
        print("seam cache on 1M x 16 (%d MB): level 3 %.2f ms per call, "
              "level 2 %.2f ms" % ((12 * NZ + 4 * M + 8 * N) >> 20, ms3, ms2))
        assert np.array_equal(y2, y3c)
    finally:
        S.seam_cache(0)
    S.csr_free(A)
