#!/usr/bin/env python3
"""Regenerate the golden fixtures from the REFERENCE's own CPU code.

Run in the build container (needs /root/reference and gcc):

    oracle/build_ref.sh && python tests/golden/make_golden.py

Writes, next to this file:
  *.mtx              small Matrix Market inputs (authored here, deterministic)
  *.ref.txt          what the reference produced for them: CSR arrays, both
                     HLL layouts, the glibc-rand x vector, serial/OpenMP y
                     (oracle/ref_harness.c `dump`, strict-IEEE build)
  err_*.mtx          inputs the reference loader rejects, with the errno it
                     returned recorded in errors.ref.txt
  synth_*.ref.txt    reference serial CSR/HLL results on the synthetic
                     families of include/spmv_synth.h (`synth` command)

The fixtures are data (inputs + expected outputs).  The GPU box never sees
the reference: tests compare the oracle restatement, the host library and the
HIP kernels against these files.
"""
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_strict")


def write(name, text):
    with open(os.path.join(HERE, name), "w") as f:
        f.write(text)


def banner(field="real", sym="general", obj="matrix", fmt="coordinate"):
    return "%%%%MatrixMarket %s %s %s %s\n" % (obj, fmt, field, sym)


def coo_text(entries, pattern=False):
    out = []
    for e in entries:
        if pattern:
            out.append("%d %d\n" % (e[0], e[1]))
        else:
            out.append("%d %d %s\n" % (e[0], e[1], repr(float(e[2]))))
    return "".join(out)


def make_inputs():
    rng = random.Random(20251003)
    files = {}

    # SURVEY 8(c) known-answer inputs
    files["gen.mtx"] = banner() + "4 5 7\n" + coo_text(
        [(1, 1, 1.5), (1, 4, -2), (2, 2, 3), (3, 1, 4), (3, 3, 5), (3, 5, 6),
         (4, 4, 7.25)])
    files["sym.mtx"] = banner(sym="symmetric") + "3 3 4\n" + coo_text(
        [(1, 1, 2), (2, 1, -1), (3, 2, -1), (3, 3, 2)])
    files["pat.mtx"] = banner("pattern", "symmetric") + "3 3 3\n" + coo_text(
        [(1, 1), (2, 1), (3, 3)], pattern=True)

    # config 1 stand-in: 9x9, 49 entries, real general (cage4 itself is not
    # available offline); unsorted file order, a few repeated coordinates.
    ent = []
    cells = [(i, j) for i in range(1, 10) for j in range(1, 10)]
    rng.shuffle(cells)
    for (i, j) in cells[:46]:
        ent.append((i, j, round(rng.uniform(-1, 1), 6)))
    ent += [(cells[0][0], cells[0][1], 0.25), (5, 5, -0.125), (5, 5, 0.5)]
    rng.shuffle(ent)
    assert len(ent) == 49
    files["cage4_like.mtx"] = (banner() + "% stand-in for cage4 (9x9, 49 nnz)\n"
                               "9 9 49\n" + coo_text(ent))

    # two hack blocks, tail block of 8 rows with no entries at all
    files["tail40.mtx"] = banner() + "40 40 3\n" + coo_text(
        [(1, 1, 1.0), (2, 40, -3.5), (31, 7, 0.5)])

    # ragged 100x80: empty rows, long row, unsorted, duplicates, tail block
    ent = []
    for i in range(1, 101):
        if i % 7 == 0:
            continue
        ln = 37 if i == 50 else rng.randint(1, 9)
        for _ in range(ln):
            ent.append((i, rng.randint(1, 80), round(rng.uniform(-2, 2), 5)))
    rng.shuffle(ent)
    files["ragged100.mtx"] = (banner() + "%comment one\n%comment two\n"
                              + "100 80 " + str(len(ent)) + "\n"
                              + coo_text(ent))

    # symmetric with diagonal + strictly lower entries, 70 rows (3 blocks)
    ent = []
    for i in range(1, 71):
        ent.append((i, i, round(rng.uniform(1, 2), 4)))
        for _ in range(rng.randint(0, 4)):
            j = rng.randint(1, i)
            ent.append((i, j, round(rng.uniform(-1, 1), 4)))
    rng.shuffle(ent)
    files["sym70.mtx"] = (banner(sym="symmetric") + "70 70 %d\n" % len(ent)
                          + coo_text(ent))

    # dc1 class in miniature (reference scripts/download-matrices.py:7-38):
    # 96 x 96 (3 hack blocks), one row holding EVERY column, one column
    # present in most rows, short rows otherwise, empty rows, file order
    # shuffled, a repeated coordinate
    ent = []
    for i in range(1, 97):
        if i in (5, 64, 65, 66):
            continue                      # empty rows
        if i == 41:
            for j in range(1, 97):        # the hub row
                ent.append((i, j, round(rng.uniform(-1, 1), 5)))
            continue
        for _ in range(rng.randint(1, 4)):
            ent.append((i, rng.randint(1, 96), round(rng.uniform(-2, 2), 5)))
        if i % 10 != 3:
            ent.append((i, 8, round(rng.uniform(-1, 1), 5)))   # the hub column
    ent.append((41, 17, 0.5))             # repeated coordinate in the hub row
    rng.shuffle(ent)
    files["hub96.mtx"] = (banner() + "% hub row 41, hub column 8\n"
                          + "96 96 %d\n" % len(ent) + coo_text(ent))

    # skew-symmetric / hermitian are NOT mirrored by the reference loader
    files["skew.mtx"] = banner(sym="skew-symmetric") + "3 3 2\n" + coo_text(
        [(2, 1, 1.5), (3, 1, -2.5)])
    files["herm.mtx"] = banner(sym="hermitian") + "3 3 2\n" + coo_text(
        [(1, 1, 1.0), (3, 2, 4.0)])
    files["patgen.mtx"] = banner("pattern") + "5 6 6\n" + coo_text(
        [(1, 6), (2, 1), (2, 2), (4, 3), (5, 5), (5, 1)], pattern=True)

    # banner case-insensitivity, blank line before the size line, free-form
    # whitespace (fscanf does not care about line structure), exponents
    files["freeform.mtx"] = (
        "%%MatrixMarket MATRIX Coordinate REAL General\n"
        "% a comment\n\n"
        "  3   4   5  \n"
        "1 1 1e-3   1 2 -.5\n"
        "2 4\n 1E+2\n"
        "3 3 +7.0e0 3 1 0x1.8p1\n")

    # inputs the loader rejects
    files["err_complex.mtx"] = banner("complex") + "2 2 1\n1 1 1.0 0.0\n"
    files["err_integer.mtx"] = banner("integer") + "2 2 1\n1 1 3\n"
    files["err_array.mtx"] = ("%%MatrixMarket matrix array real general\n"
                              "2 2\n1\n2\n3\n4\n")
    files["err_banner.mtx"] = "%MatrixMarket matrix coordinate real general\n1 1 1\n1 1 1\n"
    files["err_vector.mtx"] = "%%MatrixMarket vector coordinate real general\n1 1 1\n1 1 1\n"
    files["err_short_banner.mtx"] = "%%MatrixMarket matrix coordinate real\n1 1 1\n1 1 1\n"
    files["err_badsym.mtx"] = banner(sym="diagonal") + "1 1 1\n1 1 1\n"
    files["err_range_row.mtx"] = banner() + "2 2 2\n1 1 1.0\n3 1 2.0\n"
    files["err_range_col.mtx"] = banner() + "2 2 1\n1 0 1.0\n"
    files["err_short.mtx"] = banner() + "3 3 4\n1 1 1.0\n2 2 2.0\n"
    files["err_garbage.mtx"] = banner() + "2 2 2\n1 1 1.0\nx y z\n"
    files["err_nosize.mtx"] = banner() + "% only comments\n"
    files["err_empty.mtx"] = ""
    for k, v in files.items():
        write(k, v)
    return sorted(files)


SYNTH = {
    # name: kind M N K W seed xseed nsample
    "synth_banded": (0, 5000, 5000, 16, 0, 42, 7, 5000),
    "synth_random": (1, 4100, 4100, 32, 512, 42, 7, 4100),
    "synth_random_wide": (1, 3000, 3000, 32, 6000, 42, 7, 3000),
    "synth_ragged": (2, 2077, 2500, 32, 256, 42, 7, 2077),
    "synth_kkt": (3, 3000, 3000, 16, 3000, 42, 7, 3000),
    "synth_stencil27": (4, 3500, 3500, 27, 0, 42, 7, 3500),   # 15x15x16 grid, ragged top
    "synth_stencil7": (4, 4096, 4096, 7, 16, 42, 7, 4096),
    # reference matrix classes with very short rows + a heavy tail
    # (scripts/download-matrices.py:7-38: webbase-1M / amazon0302 / roadNet-PA; dc1)
    "synth_powerlaw": (5, 6000, 6000, 3, 12000, 42, 7, 6000),
    "synth_powerlaw_k8": (5, 3000, 3500, 8, 1024, 42, 7, 3000),
    "synth_hub": (6, 4000, 4000, 6, 256, 42, 7, 4000),   # hub row = all 4000 columns
}


def run(args):
    return subprocess.run([REF] + [str(a) for a in args], check=True,
                          capture_output=True, text=True).stdout


def main():
    if not os.path.exists(REF):
        sys.exit("build oracle/_ref first: oracle/build_ref.sh")
    names = make_inputs()
    errs = []
    for n in names:
        path = os.path.join(HERE, n)
        if n.startswith("err_"):
            errs.append("%s %s" % (n, run(["err", path]).split()[2]))
        else:
            write(n[:-4] + ".ref.txt", run(["dump", path]))
    errs.append("err_missing_file.mtx %s" % run(
        ["err", os.path.join(HERE, "does_not_exist.mtx")]).split()[2])
    write("errors.ref.txt", "\n".join(errs) + "\n")
    for name, spec in SYNTH.items():
        write(name + ".ref.txt",
              "spec %d %s\n" % (len(spec), " ".join(map(str, spec)))
              + run(["synth"] + list(spec)))
    print("wrote", len(names), "inputs,", len(SYNTH), "synthetic goldens")


if __name__ == "__main__":
    main()
