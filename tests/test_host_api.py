"""Host C API (loader, HLL converter, CPU benches, logger, partitions) against
the reference-built goldens and the oracle.  CPU only."""
import ctypes as C
import errno
import os
import subprocess

import numpy as np
import pytest

import _golden as G
import _oracle as O
import spmv_scpa_amd as S


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


REF_ABI = ["set_csr_warps_per_block", "csr_spmv_cuda_thread_row",
           "csr_spmv_cuda_warp_row", "csr_spmv_cuda_halfwarp_row",
           "csr_spmv_cuda_block_row", "csr_spmv_cuda_halfwarp_row_text",
           "set_hll_warps_per_block", "hll_spmv_cuda_threads_row_major",
           "hll_spmv_cuda_threads_col_major", "hll_spmv_cuda_warp_block",
           "hll_spmv_cuda_halfwarp_row"]


def test_library_exports_every_declared_symbol():
    names = S.declared_symbols()
    assert len(names) >= 90
    for n in REF_ABI + ["io_load_csr", "csr_to_hll", "spmv_hll_launch",
                        "csr_spmv_hip_subwave_row",
                        "hll_spmv_hip_threads_col_major"]:
        assert n in names, n
    assert S.check_symbols()
    assert "gfx950" in S.version()


@pytest.mark.parametrize("name", G.MTX_CASES)
def test_io_load_csr_matches_reference(name):
    ref = G.load_ref(name)
    A = S.io_load_csr(G.mtx_path(name))
    a = A.contents
    assert [a.M, a.N, a.NZ] == list(ref["shape"])
    assert a.name.decode() == ref["name"]
    IRP, JA, AS = S.csr_arrays(A)
    assert np.array_equal(IRP, ref["IRP"])
    assert np.array_equal(JA, ref["JA"])
    assert np.array_equal(bits(AS), bits(ref["AS"]))
    for arr in (a.IRP, a.JA, a.AS):  # 64 B alignment (reference utils.h:14)
        assert C.cast(arr, C.c_void_p).value % 64 == 0
    S.csr_free(A)


def test_io_load_csr_error_codes_match_reference():
    for fname, code in G.load_errors().items():
        path = os.path.join(G.GOLDEN, fname.replace("err_missing_file",
                                                    "does_not_exist"))
        with pytest.raises(OSError) as ei:
            S.io_load_csr(path)
        assert ei.value.errno == -code, (fname, ei.value.errno, code)


def test_matrix_name_extraction():
    assert S.extract_matrix_name("/a/b/cage4.mtx") == "cage4"
    assert S.extract_matrix_name("plain") == "plain"
    assert S.extract_matrix_name(".mtx") == ".mtx"
    long = "x" * 100 + ".mtx"
    assert S.extract_matrix_name(long) == "x" * 63 == O.matrix_name(long)


@pytest.mark.parametrize("name", G.MTX_CASES)
@pytest.mark.parametrize("col_major", [False, True])
def test_csr_to_hll_matches_reference(name, col_major):
    ref = G.load_ref(name)
    A = S.io_load_csr(G.mtx_path(name))
    H = S.csr_to_hll(A, col_major)
    hdr, blocks = G.hll_blocks(ref, "hll_col" if col_major else "hll_row")
    h = H.contents
    assert [h.M, h.N, h.NZ, h.hack_size, h.num_blocks] == list(hdr)
    assert h.name == A.contents.name
    got = S.hll_blocks(H)
    assert len(got) == len(blocks)
    slots = 0
    for (gm, gn, gnz, gmax, gja, gas), (rm, rn, rnz, rmax, rja, ras) in zip(
            got, blocks):
        assert (gm, gn, gnz, gmax) == (rm, rn, rnz, rmax)
        assert np.array_equal(gja, rja)
        assert np.array_equal(bits(gas), bits(ras))
        slots += gm * gmax
    assert S.hll_num_slots(H) == slots
    assert S._lib.hll_is_contiguous(H) == 1  # slab-backed
    S.hll_free(H)
    S.csr_free(A)


@pytest.mark.parametrize("name", G.MTX_CASES)
def test_cpu_benchmarks_match_reference_bits(name):
    ref = G.load_ref(name)
    A = S.io_load_csr(G.mtx_path(name))
    x = S.vec_random(A.contents.N)
    assert np.array_equal(bits(x), bits(ref["x"]))
    y, ms, gf = S.bench_csr_serial(A, x)
    assert np.array_equal(bits(y), bits(ref["y_csr_serial"]))
    assert ms >= 0 and gf == S.compute_gflops(ms, A.contents.NZ)
    Hr, Hc = S.csr_to_hll(A, False), S.csr_to_hll(A, True)
    assert np.array_equal(bits(S.bench_hll_serial(Hr, x)[0]),
                          bits(ref["y_hll_serial"]))
    assert np.array_equal(bits(S.bench_hll_serial(Hc, x, col_major=True)[0]),
                          bits(ref["y_hll_serial"]))
    yg, _, _, nm, thr = S.bench_csr_omp_guided(A, x, 2)
    assert nm == ref["omp_names"][0] == "omp_guided" and thr == 2
    assert np.array_equal(bits(yg), bits(ref["y_csr_omp_guided"]))
    yn, _, _, nm, thr = S.bench_csr_omp_nnz_balancing(A, x, 2)
    assert nm == ref["omp_names"][1] == "omp_nnz"
    assert thr == int(ref["omp_nnz_threads"][0])
    assert np.array_equal(bits(yn), bits(ref["y_csr_omp_nnz"]))
    yh, _, _, nm, _ = S.bench_hll_omp(Hr, x, 2)
    assert nm == ref["omp_names"][2]
    assert np.array_equal(bits(yh), bits(ref["y_hll_omp"]))
    assert S.validation_vec_result(y, yh) == 0
    assert S.max_rel_err(y, yh) == 0.0
    S.hll_free(Hr)
    S.hll_free(Hc)
    S.csr_free(A)


def test_validation_and_rel_err_semantics():
    a = np.array([1.0, 2.0, 3.0])
    assert S.validation_vec_result(a, a + 0.05) == 0      # L2 = 0.087
    assert S.validation_vec_result(a, a + 0.06) == -1     # L2 = 0.104
    assert S.validation_vec_result(a, a[:2]) == -1
    assert S.max_rel_err(a, a[:2]) == -1.0
    assert abs(S.max_rel_err(a, a * (1 + 1e-9)) - 1e-9) < 1e-12
    # cancelling row: |y| tiny, row scale large -> judged against the scale
    e = S.max_rel_err(np.array([1e-20]), np.array([3e-17]),
                      scale=np.array([1.0]))
    assert e < 1e-13
    assert S.compute_gflops(2.0, 7) == O.gflops(2.0, 7)
    assert S.compute_gflops(0.0, 7) == 0.0


@pytest.mark.parametrize("name", G.SYNTH_CASES)
def test_csr_generate_matches_oracle_and_reference(name):
    ref = G.load_ref(name)
    kind, M, N, K, W, seed, xseed, _ = [int(v) for v in ref["spec"]]
    A = S.csr_generate(kind, M, N, K, W, 0, seed)
    IRP, JA, AS = S.csr_arrays(A)
    oI, oJ, oA = O.synth_csr(kind, M, N, K, W, seed)
    assert np.array_equal(IRP, oI) and np.array_equal(JA, oJ)
    assert np.array_equal(bits(AS), bits(oA))
    x = S.vec_synth(N, xseed)
    assert np.array_equal(bits(x), bits(O.synth_x(xseed, 0, N)))
    y = S.bench_csr_serial(A, x)[0]
    st = int(ref["stride"][0])
    assert np.array_equal(bits(y[::st]), bits(ref["y_csr_serial_sample"]))
    # a shard generated with row0 equals the same rows of the whole matrix
    r0 = (M // 3) // 32 * 32
    B = S.csr_generate(kind, M - r0, N, K, W, r0, seed)
    bI, bJ, bA = S.csr_arrays(B)
    assert np.array_equal(bJ, JA[IRP[r0]:]) and np.array_equal(bA, AS[IRP[r0]:])
    Sl = S.csr_row_slice(A, r0, M)
    sI, sJ, sA = S.csr_arrays(Sl)
    assert np.array_equal(sI, bI) and np.array_equal(sJ, bJ)
    for p in (A, B, Sl):
        S.csr_free(p)


def test_partitions():
    r = G.load_ref("ragged100")
    A = S.io_load_csr(G.mtx_path("ragged100"))
    IRP = r["IRP"].astype(np.int32)
    for parts in (1, 2, 3, 8, 40):
        got = S.partition_rows_nnz(A, parts)
        assert np.array_equal(got, O.partition_rows(IRP, parts)), parts
        assert got[0] == 0 and got[-1] == 100
    S.csr_free(A)
    ev = S.partition_rows_even(10_000_000, 8, 32)
    assert list(np.diff(ev)) == [1250016] * 7 + [10_000_000 - 7 * 1250016]
    assert all(v % 32 == 0 for v in ev[:-1])
    ev = S.partition_rows_even(100, 8, 32)
    assert list(ev) == [0, 32, 64, 96, 100, 100, 100, 100, 100]
    ev = S.partition_rows_even(80_000_000, 8, 32)
    assert list(np.diff(ev)) == [10_000_000] * 8


def test_csv_logger_schema_is_byte_exact(tmp_path):
    d = str(tmp_path)
    A = S.io_load_csr(G.mtx_path("tail40"))
    H = S.csr_to_hll(A, False)
    assert S._lib.logger_init(d.encode()) == 0
    b = S.Bench(1.5, 0.25, S.Vec(0, None))
    S._lib.log_csr_serial_benchmark(A, b)
    S._lib.log_hll_serial_benchmark(H, b)
    bo = S.BenchOmp(b, b"omp_guided", 8)
    S._lib.log_csr_omp_benchmark(A, bo)
    S._lib.log_hll_omp_benchmark(H, bo)
    bh = S.BenchHip(b, 4)
    S._lib.log_csr_hip_benchmark(A, bh, 2)
    S._lib.log_hll_hip_benchmark(H, bh, 1)
    S._lib.logger_close()
    # re-open: append, no second header (reference logger.c:21-51)
    assert S._lib.logger_init(d.encode()) == 0
    S._lib.log_csr_serial_benchmark(A, b)
    S._lib.logger_close()
    assert open(os.path.join(d, "serial.csv")).read() == (
        "matrix,format,rows,cols,nnz,num_blocks,duration_ms,gflops\n"
        "tail40,CSR,40,40,3,,1.500000,0.250000\n"
        "tail40,HLL,40,40,3,2,1.500000,0.250000\n"
        "tail40,CSR,40,40,3,,1.500000,0.250000\n")
    assert open(os.path.join(d, "omp.csv")).read() == (
        "matrix,format,bench,rows,cols,nnz,num_blocks,num_threads,duration_ms,"
        "gflops\n"
        "tail40,CSR,omp_guided,40,40,3,,8,1.500000,0.250000\n"
        "tail40,HLL,omp_guided,40,40,3,2,8,1.500000,0.250000\n")
    assert open(os.path.join(d, "cuda.csv")).read() == (
        "matrix,format,kernel,warps_per_block,rows,cols,nnz,num_blocks,"
        "duration_ms,gflops\n"
        "tail40,CSR,2,4,40,40,3,,1.500000,0.250000\n"
        "tail40,HLL,1,4,40,40,3,2,1.500000,0.250000\n")
    assert S._lib.logger_init(b"/nonexistent/dir") == -1
    S.hll_free(H)
    S.csr_free(A)


def test_gpu_entry_points_fail_loudly_without_a_gpu():
    if S.device_count() > 0:
        pytest.skip("a GPU is present")
    A = S.io_load_csr(G.mtx_path("gen"))
    x = S.vec_random(5)
    for k in range(S.NUM_CSR_KERNELS):
        with pytest.raises(OSError) as ei:
            S.csr_spmv_hip(A, x, kernel=k)
        assert ei.value.errno == errno.ENODEV
        with pytest.raises(OSError) as ei:
            S.bench_csr_hip(A, x, k)
        assert ei.value.errno == errno.ENODEV
    H = S.csr_to_hll(A, True)
    with pytest.raises(OSError) as ei:
        S.hll_spmv_hip(H, x, kernel=1)
    assert ei.value.errno == errno.ENODEV
    with pytest.raises(OSError):
        S.CsrDevice.upload(A)
    with pytest.raises(OSError):
        S.DevBuffer(1024)
    S.hll_free(H)
    S.csr_free(A)


DRIVER = os.path.join(S.ROOT, "spmv_scpa_amd", "bin", "spmv_scpa_amd")


def test_driver_cli_cpu_path(tmp_path):
    out = str(tmp_path)
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run([DRIVER, "-m", G.mtx_path("sym70"), "-o", out, "-d",
                        "--iters", "2"], capture_output=True, text=True,
                       env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    serial = open(os.path.join(out, "serial.csv")).read().splitlines()
    assert serial[0] == "matrix,format,rows,cols,nnz,num_blocks,duration_ms,gflops"
    assert serial[1].startswith("sym70,CSR,70,70,") and serial[2].startswith(
        "sym70,HLL,70,70,")
    omp = open(os.path.join(out, "omp.csv")).read().splitlines()
    assert len(omp) == 1 + 3 * 2  # threads 2 and 4 x {omp_nnz, omp_guided, hll}
    # errors are reported, not crashed on (the reference segfaults: main.c:79)
    r = subprocess.run([DRIVER, "-m", "/no/such.mtx", "-o", out],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "No such file" in r.stderr
    r = subprocess.run([DRIVER, "-m", G.mtx_path("err_complex"), "-o", out],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "err -22" in r.stderr
    r = subprocess.run([DRIVER, "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage:" in r.stderr
    r = subprocess.run([DRIVER], capture_output=True, text=True)
    assert r.returncode == 1
    r = subprocess.run([DRIVER, "-s", "random", "--rows", "4096", "--nnz-row",
                        "8", "--window", "256", "-o", out, "-d", "--iters", "1"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr


def _write_big_mtx(path, M, N, nnz, sym=False, pattern=False, seed=3):
    rng = np.random.default_rng(seed)
    i = rng.integers(1, M + 1, nnz)
    j = rng.integers(1, N + 1, nnz)
    if sym:
        i, j = np.maximum(i, j), np.minimum(i, j)
    v = rng.standard_normal(nnz)
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s %s\n%% big\n%d %d %d\n"
                % ("pattern" if pattern else "real",
                   "symmetric" if sym else "general", M, N, nnz))
        if pattern:
            np.savetxt(f, np.c_[i, j], fmt="%d %d")
        else:
            for a, b, c in zip(i, j, v):
                f.write("%d %d %s\n" % (a, b, repr(float(c))))


@pytest.mark.parametrize("sym,pattern", [(False, False), (True, False),
                                         (False, True)])
def test_parallel_loader_equals_oracle(tmp_path, sym, pattern):
    """>= 100k clean entries take the multi-threaded tokeniser; the result
    must equal the oracle's two-pass fscanf restatement bit for bit."""
    p = str(tmp_path / "big.mtx")
    _write_big_mtx(p, 5000, 5000, 150_000, sym, pattern)
    rc, M, N, NZ, IRP, JA, AS = O.load_mtx(p)
    assert rc == 0
    A = S.io_load_csr(p)  # just written: read into memory, not mapped
    gI, gJ, gA = S.csr_arrays(A)
    assert A.contents.NZ == NZ
    assert np.array_equal(gI, IRP) and np.array_equal(gJ, JA)
    assert np.array_equal(bits(gA), bits(AS))
    # a file at rest (mtime > 2 s ago) is mapped: same result either way
    past = os.stat(p).st_mtime - 60
    os.utime(p, (past, past))
    A2 = S.io_load_csr(p)
    mI, mJ, mA = S.csr_arrays(A2)
    assert np.array_equal(mI, IRP) and np.array_equal(mJ, JA)
    assert np.array_equal(bits(mA), bits(AS))
    S.csr_free(A2)
    # binary sidecar round trip + cached loader
    S.csr_save_bin(A, p + ".bin")
    B = S.csr_load_bin(p + ".bin")
    bI, bJ, bA = S.csr_arrays(B)
    assert np.array_equal(bI, IRP) and np.array_equal(bJ, JA)
    assert np.array_equal(bits(bA), bits(AS))
    Cc = S.io_load_csr_cached(p)
    assert Cc.contents.name == b"big" and Cc.contents.NZ == NZ
    with open(p + ".bin", "r+b") as f:  # corrupt the magic: falls back to text
        f.write(b"XXXXXXXX")
    os.utime(p + ".bin")
    D = S.io_load_csr_cached(p)
    assert D.contents.NZ == NZ
    with pytest.raises(OSError):
        S.csr_load_bin(str(tmp_path / "missing.bin"))
    for h in (A, B, Cc, D):
        S.csr_free(h)


def test_parallel_loader_falls_back_on_unclean_tokens(tmp_path):
    """Tokens fscanf would split ("7-3"), a short file and an out-of-range
    entry late in a big file: same errno / arrays as the oracle."""
    p = str(tmp_path / "odd.mtx")
    _write_big_mtx(p, 4000, 4000, 120_000)
    lines = open(p).read().splitlines()
    for name, edit, want in (
            ("split", lambda L: L[:60_000] + ["7-3 0.5"] + L[60_001:], None),
            ("short", lambda L: L[:-5], -errno.EIO),
            ("range", lambda L: L[:100_000] + ["4001 1 1.0"] + L[100_001:],
             -errno.ERANGE),
            ("garbage", lambda L: L[:90_000] + ["1 x 2.0"] + L[90_001:],
             -errno.EIO)):
        q = str(tmp_path / (name + ".mtx"))
        open(q, "w").write("\n".join(edit(lines)) + "\n")
        rc = O.load_mtx(q)[0]
        if want is not None:
            assert rc == want, name
        if rc == 0:
            A = S.io_load_csr(q)
            _, _, _, _, IRP, JA, AS = O.load_mtx(q)
            gI, gJ, gA = S.csr_arrays(A)
            assert np.array_equal(gI, IRP) and np.array_equal(gJ, JA)
            assert np.array_equal(bits(gA), bits(AS))
            S.csr_free(A)
        else:
            with pytest.raises(OSError) as ei:
                S.io_load_csr(q)
            assert ei.value.errno == -rc, (name, ei.value.errno, rc)


# ---- the .bin sidecar is a validated cache tied to its source ----
_HDR = 8 + 16 + 24 + 64  # magic, M N NZ pad, src size/sec/nsec, name[MAX_NAME]
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sidecar_case(tmp_path):
    import shutil
    p = str(tmp_path / "gen.mtx")
    shutil.copy(os.path.join(GOLD, "gen.mtx"), p)
    A = S.io_load_csr_cached(p)  # parses the text, writes the sidecar
    I, J, V = (a.copy() for a in S.csr_arrays(A))
    S.csr_free(A)
    assert os.path.exists(p + ".bin")
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]
    assert os.path.getsize(p + ".bin") == _HDR + 4 * len(I) + 12 * len(J)
    return p, I, J, V


def test_sidecar_is_used_only_for_its_own_source(tmp_path):
    p, I, J, V = _sidecar_case(tmp_path)
    off_as = _HDR + 4 * len(I) + 4 * len(J)
    with open(p + ".bin", "r+b") as f:  # a VALID sidecar with another value:
        f.seek(off_as)                  # proves the cache is what is read
        f.write(np.float64(123.25).tobytes())
    A = S.io_load_csr_cached(p)
    assert S.csr_arrays(A)[2][0] == 123.25
    S.csr_free(A)
    # same size, same second, other nanosecond: stale -> re-parsed, replaced
    st = os.stat(p)
    txt = open(p).read().replace("1.5", "2.5")
    open(p, "w").write(txt)
    sec = st.st_mtime_ns // 10**9
    os.utime(p, ns=(st.st_atime_ns, sec * 10**9 + (st.st_mtime_ns + 7) % 10**9))
    assert os.stat(p).st_size == st.st_size
    A = S.io_load_csr_cached(p)
    assert S.csr_arrays(A)[2][0] == 2.5
    S.csr_free(A)
    B = S.csr_load_bin(p + ".bin")  # the replaced sidecar holds the new parse
    assert S.csr_arrays(B)[2][0] == 2.5
    S.csr_free(B)
    # a sidecar saved without a source (csr_save_bin) or from another file
    # is never trusted by the cached loader
    other = str(tmp_path / "sym.mtx")
    import shutil
    shutil.copy(os.path.join(GOLD, "sym.mtx"), other)
    C2 = S.io_load_csr(other)
    S.csr_save_bin(C2, p + ".bin")
    S.csr_free(C2)
    A = S.io_load_csr_cached(p)
    assert A.contents.M == 4 and S.csr_arrays(A)[2][0] == 2.5
    S.csr_free(A)


@pytest.mark.parametrize("what", ["ja_range", "ja_negative", "irp_order",
                                  "irp_end", "truncated", "extra", "magic"])
def test_corrupt_sidecar_never_reaches_the_kernels(tmp_path, what):
    import errno
    p, I, J, V = _sidecar_case(tmp_path)
    b = p + ".bin"
    with open(b, "r+b") as f:
        if what == "ja_range":
            f.seek(_HDR + 4 * len(I) + 4 * 2)
            f.write(np.int32(5).tobytes())  # N = 5: first invalid column
        elif what == "ja_negative":
            f.seek(_HDR + 4 * len(I))
            f.write(np.int32(-1).tobytes())
        elif what == "irp_order":
            f.seek(_HDR + 4 * 1)
            f.write(np.int32(4).tobytes())  # IRP = 0 4 3 6 7
        elif what == "irp_end":
            f.seek(_HDR + 4 * (len(I) - 1))
            f.write(np.int32(6).tobytes())
        elif what == "truncated":
            f.truncate(os.path.getsize(b) - 8)
        elif what == "extra":
            f.seek(0, 2)
            f.write(b"\0" * 8)
        elif what == "magic":
            f.write(b"SPMVCSR1")  # round 1's format is refused, not misread
    want = {"truncated": errno.EIO, "extra": errno.EIO,
            "magic": errno.EINVAL}.get(what, errno.EILSEQ)
    with pytest.raises(OSError) as e:
        S.csr_load_bin(b)
    assert e.value.errno == want
    # the cached loader falls back to the text and repairs the sidecar
    A = S.io_load_csr_cached(p)
    gI, gJ, gV = S.csr_arrays(A)
    assert np.array_equal(gI, I) and np.array_equal(gJ, J)
    assert np.array_equal(bits(gV), bits(V))
    S.csr_free(A)
    B = S.csr_load_bin(b)
    assert np.array_equal(S.csr_arrays(B)[1], J)
    S.csr_free(B)


def test_fast_decimal_path_is_bit_identical_to_the_reference_scanf(tmp_path):
    """The parallel tokeniser converts plain decimals itself (Clinger's fast
    path: <= 15 significant digits over 10^k, one correctly rounded division)
    and leaves everything else to strtod.  Every kind of token, cycled over a
    file big enough for the parallel path, against the oracle's fscanf
    restatement bit for bit."""
    toks = ["0.1", "-0.3", "123456789012345", "0.000123", "1.000",
            "999999999999999", "0.30000000000000004", "5.", "+.5", "-0.0",
            "00012.5000", "1e-3", "-2.5E+2", "0.1234567890123456",
            "1234567890123456", "0.0000000000000000000001",
            "0.00000000000000000000001", "3", "-7", "0x1.8p1", "inf"]
    M = N = 400
    n = 120_000
    rng = np.random.default_rng(3)
    i = rng.integers(1, M + 1, n)
    j = rng.integers(1, N + 1, n)
    p = str(tmp_path / "tok.mtx")
    with open(p, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n"
                % (M, N, n))
        f.write("".join("%d %d %s\n" % (a, b, toks[k % len(toks)])
                        for k, (a, b) in enumerate(zip(i, j))))
    rc, oM, oN, oNZ, IRP, JA, AS = O.load_mtx(p)
    assert rc == 0 and oNZ == n
    A = S.io_load_csr(p)
    gI, gJ, gA = S.csr_arrays(A)
    assert np.array_equal(gI, IRP) and np.array_equal(gJ, JA)
    assert np.array_equal(bits(gA), bits(AS))
    assert np.isinf(gA).any() and (gA == 0.1).any() and (gA == 3.0).any()
    S.csr_free(A)
