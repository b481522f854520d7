"""tools/run_all.py (batch runner over matrices x GPU counts,
CSVs the reference's scripts/plots.py can read).  CPU: the real driver on two
golden .mtx (no GPU here -> the GPU grid is skipped, the files still carry
the reference's headers), and a recording stand-in for the executable that
shows which driver invocations a GPU-count sweep makes."""
import csv
import importlib.util
import os
import stat
import statistics
import sys

import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_runner():
    spec = importlib.util.spec_from_file_location(
        "run_all", os.path.join(ROOT, "tools", "run_all.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_batch_run_writes_csvs_with_the_reference_columns(tmp_path, capsys):
    R = load_runner()
    mdir, res = tmp_path / "mtx", tmp_path / "res"
    mdir.mkdir()
    for name in ("gen", "sym70"):
        os.symlink(G.mtx_path(name), mdir / (name + ".mtx"))
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    rc = R.main(["-m", str(mdir), "-res", str(res), "-i", "2",
                 "--gpus", "1,2", "--debug"])
    assert rc == 0
    out = capsys.readouterr().out
    for fn, cols in (("serial.csv", R.SERIAL_COLS), ("omp.csv", R.OMP_COLS),
                     ("cuda.csv", R.CUDA_COLS)):
        rows = list(csv.DictReader(open(res / fn)))
        assert csv.DictReader(open(res / fn)).fieldnames == cols, fn
        if fn != "cuda.csv":  # no GPU in this container: header only
            assert {r["matrix"] for r in rows} == {"gen", "sym70"}, fn
            for r in rows:
                float(r["duration_ms"]), float(r["gflops"]), int(r["nnz"])
    ser = list(csv.DictReader(open(res / "serial.csv")))
    assert len(ser) == 2 * 2 * 2  # matrices x iterations x {CSR, HLL}
    # the printed medians are the ones plots.py would compute
    want = statistics.median(float(r["gflops"]) for r in ser
                             if (r["matrix"], r["format"]) == ("gen", "CSR"))
    med = R.medians(str(res / "serial.csv"), ["matrix", "format"])
    assert med[("gen", "CSR")][1] == want and med[("gen", "CSR")][2] == 2
    assert "serial medians" in out and "OpenMP medians" in out


def test_gpu_count_sweep_passes_g_and_reruns_only_the_multi_gpu_step(tmp_path):
    R = load_runner()
    log = tmp_path / "calls.txt"
    exe = tmp_path / "fake_driver.sh"
    exe.write_text("#!/bin/sh\necho \"$@\" >> %s\n" % log)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    mdir = tmp_path / "mtx"
    mdir.mkdir()
    os.symlink(G.mtx_path("gen"), mdir / "gen.mtx")
    rc = R.main(["-exe", str(exe), "-m", str(mdir), "-res", str(tmp_path / "r"),
                 "-i", "2", "--gpus", "8,1,4,2,16", "--assume-gpus", "8",
                 "--no-cpu"])
    assert rc == 0
    calls = [c.split() for c in open(log).read().splitlines()]
    assert len(calls) == 2 * 4  # iterations x counts 1,2,4,8 (16 skipped)
    for it in range(2):
        block = calls[4 * it:4 * it + 4]
        assert [c[c.index("-g") + 1] for c in block] == ["1", "2", "4", "8"]
        assert "--only-multi-gpu" not in block[0]  # full grid once
        assert all("--only-multi-gpu" in c for c in block[1:])
        assert all("--no-cpu" in c and "-m" in c for c in block)
    # without --gpus the invocation is the reference runner's
    os.remove(log)
    R.main(["-exe", str(exe), "-m", str(mdir), "-res", str(tmp_path / "r"),
            "-i", "1"])
    (only,) = [c.split() for c in open(log).read().splitlines()]
    assert "-g" not in only and only[:2] == ["-m", str(mdir / "gen.mtx")]


def test_partition_option_reaches_the_driver(tmp_path):
    """run_all.py --gpus N --partition nnz: the driver gets `--partition nnz
    --ragged-exchange <x>` on every invocation; the default stays silent"""
    R = load_runner()
    log = tmp_path / "calls.txt"
    exe = tmp_path / "fake_driver.sh"
    exe.write_text("#!/bin/sh\necho \"$@\" >> %s\n" % log)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    mdir = tmp_path / "mtx"
    mdir.mkdir()
    os.symlink(G.mtx_path("gen"), mdir / "gen.mtx")
    rc = R.main(["-exe", str(exe), "-m", str(mdir), "-res", str(tmp_path / "r"),
                 "-i", "1", "--gpus", "1,2", "--assume-gpus", "2",
                 "--partition", "nnz", "--ragged-exchange", "bcast"])
    assert rc == 0
    calls = [c.split() for c in open(log).read().splitlines()]
    assert len(calls) == 2
    for c in calls:
        assert c[c.index("--partition") + 1] == "nnz"
        assert c[c.index("--ragged-exchange") + 1] == "bcast"
