"""Pin the CPU oracle (oracle/spmv_oracle.c) to the reference's own outputs.

Every expected array under tests/golden/*.ref.txt was produced by the
reference's C code (oracle/_ref/ref_strict; see tests/golden/make_golden.py).
Bit-exact comparisons throughout: integers equal, doubles equal as bits.
"""
import numpy as np
import pytest

import _golden as G
import _oracle as O


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("name", G.MTX_CASES)
def test_loader_matches_reference(name):
    ref = G.load_ref(name)
    rc, M, N, NZ, IRP, JA, AS = O.load_mtx(G.mtx_path(name))
    assert rc == 0
    assert [M, N, NZ] == list(ref["shape"])
    assert np.array_equal(IRP, ref["IRP"])
    assert np.array_equal(JA, ref["JA"])
    assert np.array_equal(bits(AS), bits(ref["AS"]))
    assert O.matrix_name(G.mtx_path(name)) == ref["name"]


def test_loader_error_codes_match_reference():
    errs = G.load_errors()
    assert len(errs) >= 14
    for fname, code in errs.items():
        rc = O.load_mtx(G.GOLDEN + "/" + fname.replace(
            "err_missing_file", "does_not_exist"))[0]
        assert rc == code, (fname, rc, code)


@pytest.mark.parametrize("name", G.MTX_CASES)
@pytest.mark.parametrize("col_major", [False, True])
def test_hll_conversion_matches_reference(name, col_major):
    ref = G.load_ref(name)
    IRP, JA, AS = (ref["IRP"].astype(np.int32), ref["JA"].astype(np.int32),
                   ref["AS"])
    hdr, blocks = G.hll_blocks(ref, "hll_col" if col_major else "hll_row")
    off, maxnz, blknz, HJA, HAS = O.csr_to_hll(IRP, JA, AS, col_major)
    M = len(IRP) - 1
    assert int(hdr[4]) == len(maxnz) == (M + 31) // 32
    assert int(hdr[3]) == 32
    for b, (bm, bn, bnz, bmax, bja, bas) in enumerate(blocks):
        assert bm == min(32, M - 32 * b)
        assert bmax == maxnz[b] and bnz == blknz[b]
        assert off[b + 1] - off[b] == bm * bmax
        assert np.array_equal(HJA[off[b]:off[b + 1]], bja)
        assert np.array_equal(bits(HAS[off[b]:off[b + 1]]), bits(bas))


@pytest.mark.parametrize("name", G.MTX_CASES)
def test_serial_spmv_matches_reference_bits(name):
    ref = G.load_ref(name)
    IRP, JA, AS = (ref["IRP"].astype(np.int32), ref["JA"].astype(np.int32),
                   ref["AS"])
    N = int(ref["shape"][1])
    x = O.rand_x(N)
    assert np.array_equal(bits(x), bits(ref["x"]))  # glibc rand(), seed 1
    y = O.csr_spmv(IRP, JA, AS, x)
    assert np.array_equal(bits(y), bits(ref["y_csr_serial"]))
    for cm in (False, True):
        off, maxnz, _, HJA, HAS = O.csr_to_hll(IRP, JA, AS, cm)
        yh = O.hll_spmv(len(IRP) - 1, cm, off, maxnz, HJA, HAS, x)
        assert np.array_equal(bits(yh), bits(ref["y_hll_serial"]))
    assert np.array_equal(bits(O.csr_spmv_omp(IRP, JA, AS, x, 2)),
                          bits(ref["y_csr_omp_guided"]))
    assert O.validate(y, ref["y_hll_serial"]) == int(ref["validate"][0]) == 0


def test_known_answers_from_survey():
    """first values of x and the gen/sym/pat results."""
    x = O.rand_x(3)
    assert x[0] == 0.84018771715470952 and x[1] == 0.39438292681909304
    assert x[2] == 0.78309922375860586
    ref = G.load_ref("gen")
    assert list(ref["y_csr_serial"]) == [-0.33659849122008234,
                                         1.1831487804572791,
                                         12.746131135032574,
                                         5.7886902427015317]
    assert list(G.load_ref("sym")["y_csr_serial"]) == [
        1.285992507490326, -1.6232869409133155, 1.1718155206981187]
    assert list(G.load_ref("pat")["y_csr_serial"]) == [
        1.2345706439738025, 0.84018771715470952, 0.78309922375860586]


def test_gflops_and_partition_semantics():
    ref = G.load_ref("gen")
    assert O.gflops(2.0, int(ref["shape"][2])) == ref["gflops_probe"][0]
    assert O.gflops(0.0, 10) == 0.0 and O.gflops(-1.0, 10) == 0.0
    r = G.load_ref("ragged100")
    IRP = r["IRP"].astype(np.int32)
    # reference shrinks the thread count when the greedy cut runs out of rows
    st = O.partition_rows(IRP, 2)
    assert len(st) - 1 == int(r["omp_nnz_threads"][0])
    assert st[0] == 0 and st[-1] == len(IRP) - 1
    assert all(st[i] <= st[i + 1] for i in range(len(st) - 1))


@pytest.mark.parametrize("name", G.SYNTH_CASES)
def test_synthetic_families_match_reference(name):
    ref = G.load_ref(name)
    kind, M, N, K, W, seed, xseed, _ = [int(v) for v in ref["spec"]]
    IRP, JA, AS = O.synth_csr(kind, M, N, K, W, seed)
    assert [M, N, len(JA)] == list(ref["shape"])
    x = O.synth_x(xseed, 0, N)
    y = O.csr_spmv(IRP, JA, AS, x)
    stride = int(ref["stride"][0])
    assert np.array_equal(bits(y[::stride]), bits(ref["y_csr_serial_sample"]))
    assert int(ref["hll_bit_equal"][0]) == 1
    off, maxnz, _, HJA, HAS = O.csr_to_hll(IRP, JA, AS, True)
    assert np.array_equal(bits(O.hll_spmv(M, True, off, maxnz, HJA, HAS, x)),
                          bits(y))
    # per-row regeneration agrees with the materialised matrix
    for g in (0, 1, M // 2, M - 1):
        v, ab = O.synth_row_dot(kind, M, N, K, W, 0, seed, xseed, g)
        assert v == y[g] and ab >= abs(v)
    # rows sorted ascending, columns in range
    assert JA.min() >= 0 and JA.max() < N
    for i in (0, M // 3, M - 1):
        seg = JA[IRP[i]:IRP[i + 1]]
        assert np.all(np.diff(seg) >= 0)


def test_pad_rewrite_restates_reference_upload():
    """reference cuda_hll.cu:173-195: pad -> previous valid column, or 0."""
    ref = G.load_ref("gen")
    IRP, JA, AS = (ref["IRP"].astype(np.int32), ref["JA"].astype(np.int32),
                   ref["AS"])
    off, maxnz, _, HJA, _ = O.csr_to_hll(IRP, JA, AS, False)
    fixed = O.hll_fix_pads(4, False, off, maxnz, HJA)
    assert list(fixed) == [0, 3, 3, 1, 1, 1, 0, 2, 4, 3, 3, 3]
    offc, maxc, _, HJc, _ = O.csr_to_hll(IRP, JA, AS, True)
    fixc = O.hll_fix_pads(4, True, offc, maxc, HJc)
    assert list(fixc) == [0, 1, 0, 3, 3, 1, 2, 3, 3, 1, 4, 3]
    r = G.load_ref("tail40")
    IRP = r["IRP"].astype(np.int32)
    off, maxnz, _, HJA, _ = O.csr_to_hll(IRP, r["JA"].astype(np.int32),
                                         r["AS"], True)
    fixed = O.hll_fix_pads(40, True, off, maxnz, HJA)
    assert fixed.min() >= 0
