"""The selector's ownership logic (which blocked copies are built, kept and
freed: spmv_scpa_amd/csrc/tune_blocked.h, the template engine.hip instantiates
with panels.hip's operations) compiled for the CPU with mock copies and run
under AddressSanitizer + UBSan: 20 000 seeded scenarios with failing builds,
failing timing calls and every win / lose order.  ADVICE r03: the host-side
swap / free logic of the round-2 abort's suspects, covered where a sanitizer
exists (GPU sanitizers are not available on the pool)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tune_blocked_ownership_under_asan(tmp_path):
    exe = str(tmp_path / "tune_blocked_asan")
    subprocess.run(
        ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined",
         "-fno-omit-frame-pointer", "-fno-sanitize-recover=all",
         "-I", os.path.join(ROOT, "spmv_scpa_amd", "csrc"),
         os.path.join(ROOT, "tests", "asan", "tune_blocked_asan.cc"),
         "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    r = subprocess.run([exe, "20000"], capture_output=True, text=True, env=env)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "books balance" in r.stdout and "ERROR" not in r.stderr


def test_stream_table_invariants_under_asan(tmp_path):
    """the CSR stream kernel's row-block table (stream_table.h: ranges of up
    to 2048 entries / 1024 rows, rows beyond 8192 entries cut into 2048-entry
    segments): 600 seeded row-length vectors, every entry covered by exactly
    one range, every row written exactly once, segment arithmetic as the
    kernel restates it"""
    exe = str(tmp_path / "stream_table_asan")
    # the budgets the kernel is compiled with (hip_common.h)
    import re
    hdr = open(os.path.join(ROOT, "spmv_scpa_amd", "csrc", "hip_common.h")).read()
    defs = []
    for name in ("STREAM_NNZ", "STREAM_ROWS", "STREAM_ROW_T", "STREAM_LONG_ROW",
                 "STREAM_SEG"):
        m = re.search(r"#define\s+%s\s+(\d+)" % name, hdr)
        assert m, name
        defs.append("-D%s=%s" % (name, m.group(1)))
    subprocess.run(
        ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined",
         "-fno-omit-frame-pointer", "-fno-sanitize-recover=all"] + defs +
        ["-I", os.path.join(ROOT, "spmv_scpa_amd", "csrc"),
         os.path.join(ROOT, "tests", "asan", "stream_table_asan.cc"),
         "-o", exe], check=True)
    r = subprocess.run([exe, "600"], capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "every entry covered once" in r.stdout and "ERROR" not in r.stderr
