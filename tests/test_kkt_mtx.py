"""BASELINE config 4 stand-in on the CPU: the nlpkkt160-shaped KKT file of
tools/gen_kkt_mtx.c through the product loader, against the oracle's
restatement of the reference loader (src/csr.c:31-171: file order, symmetric
mirroring) and against the matrix definition itself (tests/_kkt.py).  The
full-size run (n = 160, 8 345 600 rows) is the -m gpu test in
tests/test_gpu_config4.py; here n = 12 (serial parser) and n = 40 (1.8 M
entries: the multi-threaded tokeniser)."""
import numpy as np
import pytest

import _kkt as K
import _oracle as O
import spmv_scpa_amd as S


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("n", [12, 40])
def test_kkt_file_loads_like_the_reference_and_matches_its_definition(tmp_path, n):
    p = K.write_mtx(n, str(tmp_path / ("kkt%d.mtx" % n)))
    M, stored, nnz = K.expected_counts(n)
    G, B, n1, M2 = K.dims(n)
    assert M == M2
    rc, oM, oN, oNZ, IRP, JA, AS = O.load_mtx(p)
    assert rc == 0 and (oM, oN, oNZ) == (M, M, nnz)
    A = S.io_load_csr_cached(p)
    gI, gJ, gA = S.csr_arrays(A)
    assert (A.contents.M, A.contents.N, A.contents.NZ) == (M, M, nnz)
    assert np.array_equal(gI, IRP) and np.array_equal(gJ, JA)
    assert np.array_equal(bits(gA), bits(AS))
    Bc = S.io_load_csr_cached(p)  # second time: from the sidecar
    bI, bJ, bA = S.csr_arrays(Bc)
    assert np.array_equal(bI, IRP) and np.array_equal(bJ, JA)
    assert np.array_equal(bits(bA), bits(AS))
    # rows against the definition (independent of any parser)
    rng = np.random.default_rng(n)
    rows = np.unique(np.concatenate([[0, G - 1, G, n1 - 1, n1, M - 1],
                                     rng.integers(0, M, 300)]))
    lens = set()
    for i in rows:
        c, v = K.row(n, int(i))
        a, b = IRP[i], IRP[i + 1]
        o1, o2 = np.argsort(c, kind="stable"), np.argsort(JA[a:b], kind="stable")
        assert np.array_equal(c[o1], JA[a:b][o2]), i
        assert np.array_equal(bits(v[o1]), bits(AS[a:b][o2])), i
        lens.add(b - a)
    assert min(lens) == 2 and max(lens) == 42  # controls ... interior states
    assert np.any(AS == 0.0)  # explicit zeros are entries, not dropped
    # the matrix is symmetric: y = K x equals y = K' x
    x = O.synth_x(7, 0, M)
    y = O.csr_spmv(IRP, JA, AS, x)
    yt = np.zeros(M)
    np.add.at(yt, JA, AS * np.repeat(x, np.diff(IRP)))
    assert np.max(np.abs(y - yt)) <= 1e-12 * max(1.0, np.max(np.abs(y)))
    S.csr_free(A)
    S.csr_free(Bc)
