"""Device memory, stream and events of the SINGLE-GPU bench path, straight
from the product library's C-ABI -- so that `python bench.py` (N = 1) never
imports torch and therefore runs the library on the ROCm runtime it was built
for (/opt/rocm, HIP 7.2), not on the older copy PyTorch's wheel bundles and
maps into any process that imports it (VERDICT r05 next #5:
`config.rocm.hip_built == hip_runtime`).

Only the handful of torch calls benchlib makes at N = 1 are mirrored, with the
same names, so that benchlib.dist / benchlib.single take either module:
tensors of float64 on the current device (`empty`, `zeros`, `from_numpy`,
`data_ptr`, gather by index -> numpy), `cuda.synchronize`, `cuda.Event` pairs
on the launch stream, `cuda.current_stream().cuda_stream` (NULL: the legacy
default stream -- what torch's current stream is in a process that never
changed it).  N > 1 needs torch.distributed and takes torch itself."""
import numpy as np

import spmv_scpa_amd as S

float64 = np.float64


class _Host:
    """what `tensor.cpu()` returns"""

    def __init__(self, a):
        self.a = a

    def numpy(self):
        return self.a

    def cpu(self):
        return self


class Tensor:
    """n float64 values in HBM (spmv_dev_malloc), or a view into one"""

    def __init__(self, n, buf=None, offset=0):
        self.n = int(n)
        self.buf = buf if buf is not None else S.DevBuffer(max(self.n, 1) * 8)
        self.offset = int(offset)

    def data_ptr(self):
        return self.buf.ptr + 8 * self.offset

    def numel(self):
        return self.n

    def cpu(self):
        out = np.empty(self.n)
        if self.n:
            S._check(S._lib.spmv_copy_d2h(out.ctypes.data, self.data_ptr(),
                                          self.n * 8), "spmv_copy_d2h")
        return _Host(out)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            a, b, st = idx.indices(self.n)
            assert st == 1
            return Tensor(max(b - a, 0), self.buf, self.offset + a)
        idx = np.asarray(idx.a if isinstance(idx, _Host) else idx,
                         dtype=np.int64)
        out = np.empty(len(idx))
        one = np.empty(1)
        for k, i in enumerate(idx):  # a few hundred sampled rows
            S._check(S._lib.spmv_copy_d2h(one.ctypes.data,
                                          self.data_ptr() + 8 * int(i), 8),
                     "spmv_copy_d2h")
            out[k] = one[0]
        return _Host(out)


def empty(n, dtype=float64, device=None):
    assert dtype is float64
    return Tensor(n)


def zeros(n, dtype=float64, device=None):
    t = Tensor(n)
    S._check(S._lib.spmv_dev_memset(t.data_ptr(), 0, max(int(n), 1) * 8, None),
             "spmv_dev_memset")
    S.stream_sync()
    return t


class _FromNumpy:
    def __init__(self, a):
        self.a = np.ascontiguousarray(a, dtype=np.float64)

    def to(self, device=None):
        t = Tensor(len(self.a))
        if len(self.a):
            S._check(S._lib.spmv_copy_h2d(t.data_ptr(), self.a.ctypes.data,
                                          len(self.a) * 8), "spmv_copy_h2d")
        return t


def from_numpy(a):
    return _FromNumpy(a)


def as_tensor(a, device=None):
    return _Host(np.asarray(a))


def device(kind, index=0):
    return (kind, index)


class _Stream:
    cuda_stream = None  # NULL: the legacy default stream


class _Event:
    def __init__(self, enable_timing=True):
        self.ev = S.Event()

    def record(self):
        self.ev.record(None)

    def elapsed_time(self, stop):
        return self.ev.elapsed_ms(stop.ev)


class cuda:  # noqa: N801 - mirrors torch.cuda
    Event = _Event

    @staticmethod
    def is_available():
        return S.device_count() > 0

    @staticmethod
    def device_count():
        return S.device_count()

    @staticmethod
    def set_device(d):
        S.set_device(d)

    @staticmethod
    def synchronize():
        S.device_sync()

    @staticmethod
    def current_stream():
        return _Stream

    @staticmethod
    def empty_cache():
        return None
