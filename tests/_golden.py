"""Parser for tests/golden/*.ref.txt (format: oracle/ref_harness.c)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MTX_CASES = ["gen", "sym", "pat", "cage4_like", "tail40", "ragged100", "sym70",
             "skew", "herm", "patgen", "freeform", "hub96"]
SYNTH_CASES = ["synth_banded", "synth_random", "synth_random_wide",
               "synth_ragged", "synth_kkt", "synth_stencil27",
               "synth_stencil7", "synth_powerlaw", "synth_powerlaw_k8",
               "synth_hub"]

_INT_KEYS = ("IRP", "JA", "shape", "hdr", "validate", "omp_nnz_threads",
             "error", "stride", "hll_bit_equal", "spec")


def _is_int_key(key):
    last = key.split(".")[-1]
    return last in _INT_KEYS or last.startswith("blk") and not last.endswith("AS")


def load_ref(name):
    """-> dict key -> numpy array (int64 or float64) / str."""
    out = {}
    with open(os.path.join(GOLDEN, name + ".ref.txt")) as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            key = parts[0]
            if key in ("name", "omp_names"):
                out[key] = parts[1:] if key == "omp_names" else (
                    parts[1] if len(parts) > 1 else "")
                continue
            n = int(parts[1])
            vals = parts[2:2 + n]
            assert len(vals) == n, (name, key)
            if _is_int_key(key) and not key.endswith(".AS"):
                out[key] = np.array([int(v) for v in vals], dtype=np.int64)
            else:
                out[key] = np.array([float.fromhex(v) for v in vals],
                                    dtype=np.float64)
    return out


def load_errors():
    out = {}
    with open(os.path.join(GOLDEN, "errors.ref.txt")) as f:
        for line in f:
            if line.strip():
                k, v = line.split()
                out[k] = int(v)
    return out


def hll_blocks(ref, tag):
    """-> list of (M, N, NZ, max_NZ, JA, AS) per block."""
    hdr = ref[tag + ".hdr"]
    blocks = []
    for b in range(int(hdr[4])):
        m = ref["%s.blk%d" % (tag, b)]
        blocks.append((int(m[0]), int(m[1]), int(m[2]), int(m[3]),
                       ref["%s.blk%d.JA" % (tag, b)].astype(np.int32),
                       ref["%s.blk%d.AS" % (tag, b)]))
    return hdr, blocks


def mtx_path(name):
    return os.path.join(GOLDEN, name + ".mtx")
