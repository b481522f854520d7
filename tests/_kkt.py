"""The nlpkkt160-shaped KKT matrix of tools/gen_kkt_mtx.c restated row by row
(BASELINE config 4 stand-in: the real SuiteSparse file cannot be fetched).

    K = [H A'; A 0],  G = n^3 states, B = 6 n^2 controls, n1 = G + B, M = n1 + G

row(n, i) -> (columns, values) of row i of the FULL (mirrored) matrix, in no
particular order: what the loader must produce from the lower-triangle file,
computed without reading the file.  Test infrastructure only.
"""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "spmv_scpa_amd", "bin", "gen_kkt_mtx")
_M64 = (1 << 64) - 1


def _mix(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def value(i, j):
    """entry (i, j) = (j, i); three decimals, as the file prints it"""
    if i < j:
        i, j = j, i
    k = _mix((i << 32) | j) % 2001 - 1000
    return float("%d.%03d" % (abs(k) // 1000, abs(k) % 1000)) * (-1 if k < 0 else 1)


def dims(n):
    G, B = n ** 3, 6 * n * n
    return G, B, G + B, 2 * G + B


def control_point(n, c):
    f, r = divmod(c, n * n)
    u, v = r % n, r // n
    x, y, z = ((0, u, v), (n - 1, u, v), (u, 0, v), (u, n - 1, v), (u, v, 0),
               (u, v, n - 1))[f]
    return x + n * (y + n * z)


def _controls_at(n, g):
    """controls whose boundary point is grid point g"""
    x, y, z = g % n, g // n % n, g // (n * n)
    out = []
    for f, (hit, u, v) in enumerate(((x == 0, y, z), (x == n - 1, y, z),
                                     (y == 0, x, z), (y == n - 1, x, z),
                                     (z == 0, x, y), (z == n - 1, x, y))):
        if hit:
            out.append(f * n * n + u + n * v)
    return out


def _nbhd(n, g, fifteen):
    x, y, z = g % n, g // n % n, g // (n * n)
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                s = abs(dx) + abs(dy) + abs(dz)
                if fifteen and not (s <= 1 or s == 3):
                    continue
                if 0 <= x + dx < n and 0 <= y + dy < n and 0 <= z + dz < n:
                    yield g + dx + n * (dy + n * dz)


def row(n, i):
    G, B, n1, M = dims(n)
    cols = []
    if i < G:  # state: H row + A' part
        cols += list(_nbhd(n, i, False))
        cols += [n1 + h for h in _nbhd(n, i, True)]
    elif i < n1:  # control: diagonal + coupling
        cols += [i, n1 + control_point(n, i - G)]
    else:  # constraint row of A
        g = i - n1
        cols += list(_nbhd(n, g, True))
        cols += [G + c for c in _controls_at(n, g)]
    return np.array(cols, dtype=np.int64), np.array([value(i, j) for j in cols])


def row_dot(n, i, x):
    """(dot, sum |terms|) of row i with x"""
    c, v = row(n, i)
    t = v * x[c]
    return float(np.sum(t)), float(np.sum(np.abs(t)))


def expected_counts(n):
    """(M, stored entries, nnz after mirroring) as the generator reports"""
    out = subprocess.run([GEN, str(n), "-"], check=True, capture_output=True,
                         text=True).stdout.split()
    return int(out[0]), int(out[2]), int(out[3])


def write_mtx(n, path):
    subprocess.run([GEN, str(n), path], check=True, capture_output=True)
    return path
