"""SURVEY 8f-4 on hardware: tools/run_all.py drives the REAL driver over two
golden .mtx on the MI355X -- the counterpart of the reference's
scripts/results.py:17-28 (N process runs per matrix, CSVs appended) -- and
prints the medians its scripts/plots.py:21-53 would compute.  The CPU test
(tests/test_run_all.py) can only check headers; here the cuda.csv /
roofline.csv rows are produced by the HIP kernels, validated by the driver's
-d check against serial CSR (main.c: every variant, abs L2 <= 0.1 like the
reference, plus this build's relative bound)."""
import csv
import importlib.util
import os
import statistics

import pytest

import _golden as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_runner():
    spec = importlib.util.spec_from_file_location(
        "run_all", os.path.join(ROOT, "tools", "run_all.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_batch_runner_through_the_real_driver_on_the_gpu(tmp_path, capfd):
    R = load_runner()
    mdir, res = tmp_path / "mtx", tmp_path / "res"
    mdir.mkdir()
    names = ("gen", "sym70")
    for name in names:
        os.symlink(G.mtx_path(name), mdir / (name + ".mtx"))
    os.environ.setdefault("OMP_NUM_THREADS", "40")  # the 2..40 thread ladder
    iters = 2
    rc = R.main(["-m", str(mdir), "-res", str(res), "-i", str(iters),
                 "--gpus", "1", "--debug"])
    out = capfd.readouterr().out
    assert rc == 0, out[-3000:]

    # cuda.csv: reference schema, 27 rows per matrix and process run
    # (5 CSR + 4 HLL kernels x waves 2/4/8: main.c:258-354 of the reference)
    rd = csv.DictReader(open(res / "cuda.csv"))
    assert rd.fieldnames == R.CUDA_COLS
    rows = list(rd)
    assert len(rows) == len(names) * iters * 27
    for name in names:
        mine = [r for r in rows if r["matrix"] == name]
        assert len(mine) == iters * 27
        grid = {(r["format"], int(r["kernel"]), int(r["warps_per_block"]))
                for r in mine}
        assert grid == ({("CSR", k, w) for k in range(5) for w in (2, 4, 8)}
                        | {("HLL", k, w) for k in range(4) for w in (2, 4, 8)})
        for r in mine:
            assert float(r["duration_ms"]) > 0 and float(r["gflops"]) > 0
            assert (r["num_blocks"] == "") == (r["format"] == "CSR")

    # roofline.csv: resident timings (gpus = 1) + the `-g 1` step rows
    rd = csv.DictReader(open(res / "roofline.csv"))
    assert "gpus" in rd.fieldnames and "roofline_frac" in rd.fieldnames
    roof = list(rd)
    assert {r["gpus"] for r in roof} == {"1"}
    assert {r["matrix"] for r in roof} == set(names)
    for r in roof:
        assert float(r["duration_ms"]) > 0 and 0 <= float(r["roofline_frac"]) < 1

    # the printed medians are what plots.py computes from the same file
    med = R.medians(str(res / "cuda.csv"),
                    ["matrix", "format", "kernel", "warps_per_block"])
    key = ("gen", "HLL", "1", "4")
    want = statistics.median(
        float(r["gflops"]) for r in rows
        if (r["matrix"], r["format"], r["kernel"], r["warps_per_block"]) == key)
    assert med[key][1] == want and med[key][2] == iters
    assert "GPU medians" in out and "gen HLL 1 4" in out
    assert "steps incl. all-gather" in out
    # CPU files written in the same run keep the reference's headers
    assert csv.DictReader(open(res / "serial.csv")).fieldnames == R.SERIAL_COLS
    assert csv.DictReader(open(res / "omp.csv")).fieldnames == R.OMP_COLS
