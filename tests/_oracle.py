"""ctypes binding of oracle/liboracle.so -- the CPU checker (tests only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))

_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_int64)


def _i(a):
    return a.ctypes.data_as(_ip)


def _d(a):
    return a.ctypes.data_as(_dp)


def _l(a):
    return a.ctypes.data_as(_lp)


_lib.oracle_load_mtx.restype = C.c_int
_lib.oracle_load_mtx.argtypes = [C.c_char_p, _ip, _ip, _ip, C.POINTER(_ip),
                                 C.POINTER(_ip), C.POINTER(_dp)]
_lib.oracle_free.argtypes = [C.c_void_p]
_lib.oracle_gflops.restype = C.c_double
_lib.oracle_gflops.argtypes = [C.c_double, C.c_int]
_lib.oracle_hll_layout.restype = C.c_int64
_lib.oracle_synth_nnz.restype = C.c_int64
_lib.oracle_synth_nnz.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                  C.c_int64, C.c_uint64]
_lib.oracle_synth_csr.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                  C.c_int64, C.c_uint64, _ip, _ip, _dp]
_lib.oracle_synth_x.argtypes = [C.c_uint64, C.c_int64, C.c_int64, _dp]
_lib.oracle_synth_row_dot.restype = C.c_double
_lib.oracle_synth_row_dot.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int64, C.c_int64, C.c_uint64,
                                      C.c_uint64, C.c_int64, _dp]
_lib.oracle_time_csr_ms.restype = C.c_double
_lib.oracle_matrix_name.argtypes = [C.c_char_p, C.c_char_p]


def load_mtx(path):
    """-> (rc, M, N, NZ, IRP, JA, AS); arrays are numpy copies."""
    M, N, NZ = C.c_int(), C.c_int(), C.c_int()
    irp, ja, as_ = _ip(), _ip(), _dp()
    rc = _lib.oracle_load_mtx(path.encode(), C.byref(M), C.byref(N),
                              C.byref(NZ), C.byref(irp), C.byref(ja),
                              C.byref(as_))
    if rc:
        return rc, 0, 0, 0, None, None, None
    IRP = np.ctypeslib.as_array(irp, (M.value + 1,)).copy()
    JA = np.ctypeslib.as_array(ja, (max(NZ.value, 1),))[:NZ.value].copy()
    AS = np.ctypeslib.as_array(as_, (max(NZ.value, 1),))[:NZ.value].copy()
    for p in (irp, ja, as_):
        _lib.oracle_free(C.cast(p, C.c_void_p))
    return 0, M.value, N.value, NZ.value, IRP.astype(np.int32), \
        JA.astype(np.int32), AS


def matrix_name(path):
    buf = C.create_string_buffer(64)
    _lib.oracle_matrix_name(path.encode(), buf)
    return buf.value.decode()


def csr_spmv(IRP, JA, AS, x):
    M = len(IRP) - 1
    y = np.zeros(M)
    _lib.oracle_csr_spmv(C.c_int(M), _i(IRP), _i(JA), _d(AS), _d(x), _d(y))
    return y


def csr_abs_spmv(IRP, JA, AS, x):
    M = len(IRP) - 1
    s = np.zeros(M)
    _lib.oracle_csr_abs_spmv(C.c_int(M), _i(IRP), _i(JA), _d(AS), _d(x), _d(s))
    return s


def csr_spmv_omp(IRP, JA, AS, x, threads):
    M = len(IRP) - 1
    y = np.zeros(M)
    _lib.oracle_csr_spmv_omp(C.c_int(M), _i(IRP), _i(JA), _d(AS), _d(x), _d(y),
                             C.c_int(threads))
    return y


def partition_rows(IRP, threads):
    M = len(IRP) - 1
    t = C.c_int(threads)
    starts = np.zeros(threads + 1, dtype=np.int32)
    _lib.oracle_partition_rows(C.c_int(M), _i(IRP), C.byref(t), _i(starts))
    return starts[:t.value + 1].copy()


def hll_layout(IRP):
    M = len(IRP) - 1
    nb = (M + 31) // 32
    off = np.zeros(nb + 1, dtype=np.int64)
    maxnz = np.zeros(max(nb, 1), dtype=np.int32)
    blknz = np.zeros(max(nb, 1), dtype=np.int32)
    s = _lib.oracle_hll_layout(C.c_int(M), _i(IRP), _l(off), _i(maxnz),
                               _i(blknz))
    return int(s), off, maxnz[:nb], blknz[:nb]


def csr_to_hll(IRP, JA, AS, col_major):
    M = len(IRP) - 1
    s, off, maxnz, blknz = hll_layout(IRP)
    HJA = np.zeros(max(s, 1), dtype=np.int32)
    HAS = np.zeros(max(s, 1))
    mz = np.ascontiguousarray(maxnz) if len(maxnz) else np.zeros(1, np.int32)
    _lib.oracle_csr_to_hll(C.c_int(M), _i(IRP), _i(JA), _d(AS),
                           C.c_int(int(col_major)), _l(off), _i(mz), _i(HJA),
                           _d(HAS))
    return off, maxnz, blknz, HJA[:s], HAS[:s]


def hll_spmv(M, col_major, off, maxnz, HJA, HAS, x):
    y = np.zeros(M)
    mz = np.ascontiguousarray(maxnz) if len(maxnz) else np.zeros(1, np.int32)
    hj = HJA if len(HJA) else np.zeros(1, np.int32)
    ha = HAS if len(HAS) else np.zeros(1)
    _lib.oracle_hll_spmv(C.c_int(M), C.c_int(int(col_major)), _l(off), _i(mz),
                         _i(hj), _d(ha), _d(x), _d(y))
    return y


def hll_fix_pads(M, col_major, off, maxnz, HJA):
    out = HJA.copy() if len(HJA) else np.zeros(1, np.int32)
    mz = np.ascontiguousarray(maxnz) if len(maxnz) else np.zeros(1, np.int32)
    _lib.oracle_hll_fix_pads(C.c_int(M), C.c_int(int(col_major)), _l(off),
                             _i(mz), _i(out))
    return out[:len(HJA)]


def rand_x(n):
    x = np.zeros(n)
    _lib.oracle_rand_x(_d(x), C.c_size_t(n))
    return x


def gflops(ms, nnz):
    return _lib.oracle_gflops(ms, nnz)


def validate(a, b):
    return _lib.oracle_validate(_d(a), C.c_size_t(len(a)), _d(b),
                                C.c_size_t(len(b)))


def synth_csr(kind, M, N, K, W, seed=42, row0=0):
    nz = _lib.oracle_synth_nnz(kind, M, N, K, W, row0, seed)
    IRP = np.zeros(M + 1, dtype=np.int32)
    JA = np.zeros(max(nz, 1), dtype=np.int32)
    AS = np.zeros(max(nz, 1))
    _lib.oracle_synth_csr(kind, M, N, K, W, row0, seed, _i(IRP), _i(JA), _d(AS))
    return IRP, JA[:nz], AS[:nz]


def synth_x(seed, first, n):
    x = np.zeros(n)
    _lib.oracle_synth_x(seed, first, n, _d(x))
    return x


def synth_row_dot(kind, M, N, K, W, row0, seed, xseed, grow):
    ab = C.c_double()
    v = _lib.oracle_synth_row_dot(kind, M, N, K, W, row0, seed, xseed, grow,
                                  C.byref(ab))
    return v, ab.value


def time_csr_ms(IRP, JA, AS, x, threads, reps):
    M = len(IRP) - 1
    y = np.zeros(M)
    return _lib.oracle_time_csr_ms(C.c_int(M), _i(IRP), _i(JA), _d(AS), _d(x),
                                   _d(y), C.c_int(threads), C.c_int(reps))
