"""The multi-GPU code path on a 1-GPU box: bench.py launched through
torch.distributed.run with one rank and --force-exchange initialises RCCL
(backend "nccl"), runs the partition logic and the in-place all-gather of y
(a world of one), and must still pass its own parity spot check."""
import json
import os
import subprocess
import sys

import pytest

import spmv_scpa_amd as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [
    [], ["--chunks", "4"],
    # logical shards (the fixed-problem form of config 5): 4 matrices per
    # GPU, each all-gathered while the next one computes
    ["--shards-per-gpu", "4"],
    ["--shards-per-gpu", "2", "--kernel", "4", "--window", "0"],
    # the blocked path (sweep schedule) with an exchange: the serial
    # arrangement and two overlapped logical shards on fewer CUs are both
    # timed and the faster one kept -- either outcome is legitimate
    ["--kernel", "4", "--window", "0", "--expect-shards", "1or2"],
    # autotuned: the pick (kernel, blocked schedule, tile height) is broadcast
    ["--kernel", "-1", "--window", "65536"],
    # opt-in halo exchange (a world of one has nobody to send to)
    ["--exchange", "halo"],
])
def test_bench_through_torchrun_one_rank(extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(S.ROOT, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--rows-per-gpu",
           "320000", "--window", "4096", "--kernel", "2", "--force-exchange",
           "--kkt-n", "24",
           "--no-cpu-baseline", "--no-extras"]  # later flags override
    expect = None
    if "--expect-shards" in extra:
        i = extra.index("--expect-shards")
        expect, extra = extra[i + 1], extra[:i] + extra[i + 2:]
    cmd += extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["unit"] == "GFLOP/s"
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["frac"] > 0
    if "--shards-per-gpu" in extra:
        L = int(extra[extra.index("--shards-per-gpu") + 1])
        assert j["config"]["logical_shards_per_gpu"] == L
        assert j["config"]["rows_per_gpu"] == 320000 * L
        assert j["config"]["exchange"] == "staged"
    if "--exchange" in extra:
        assert j["config"]["exchange"] == "halo"
        assert j["config"]["halo_rows"] == 2048  # half of --window 4096
    if expect:
        # the sweep schedule: the plain arrangement is the line; the split one
        # (two logical shards beside RCCL on reserved CUs) is the optional leg
        c = j["config"]
        assert c["exchange_arrangement"].startswith("plain")
        assert c["logical_shards_per_gpu"] == 1 and c["rows_per_gpu"] == 320000
        assert c["exchange"] == "allgather"
        arr = c["arrangements"]
        assert arr["alternative_kind"] == "sweep_split", arr
        assert arr["winner"] in ("plain", "sweep_split")
        assert j["value_best"] >= j["value"] * 0.999 and j["legs_failed"] == []
        # the fixed-problem leg over 1-rank RCCL: 8 logical shards on this GPU
        st = c["strong"]
        assert st["ms_per_step"] > 0 and "8 logical shards" in st["problem"]
    # the nnz-balanced partition leg runs over the 1-rank RCCL group too
    kkt = j["config"]["partition_kkt"]
    assert kkt["nnz_balanced"]["ms_per_step"] > 0, j["legs_failed"]
    assert kkt["nnz_balanced"]["exchange"] == "p2p"
    assert j["legs_failed"] == [] and "partition_kkt" in j["legs_s"]
    # every line of a run with an exchange is printed twice: provisional
    # right after the main measurement, final at the end
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines[0]["provisional"] is True and "provisional" not in lines[-1]
    assert lines[0]["value"] == lines[-1]["value"]
    # ranks import torch (torch.distributed): the line says which runtime ran
    rocm = j["config"]["rocm"]
    assert rocm["bound"] == "torch" and rocm["runtimes_mapped"] == 1
    assert rocm["mismatch"] == (rocm["hip_built"].split(".")[:2] !=
                                rocm["hip_runtime"].split(".")[:2])


def test_the_single_gpu_line_runs_on_the_runtime_it_was_built_for():
    """`python bench.py` (what the driver runs at N = 1)
    never imports torch: the library is bound to /opt/rocm's runtime, the
    line's config.rocm shows hip_built == hip_runtime (major.minor), one
    runtime mapped, no mismatch."""
    r = subprocess.run([sys.executable, os.path.join(S.ROOT, "bench.py"),
                        "--steps", "3", "--warmup", "1", "--rows-per-gpu",
                        "640000", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    j = lines[-1]
    rocm = j["config"]["rocm"]
    assert rocm["bound"] == "system" and not rocm["shared_with_torch"], rocm
    assert rocm["hip_from"].startswith("/opt/rocm"), rocm
    assert rocm["hip_built"].split(".")[:2] == rocm["hip_runtime"].split(".")[:2]
    assert rocm["mismatch"] is False and rocm["runtimes_mapped"] == 1
    assert "rocm_mismatch" not in j
    assert j["value"] > 0 and j["rows_checked"] >= 258
    assert j["config"]["deterministic"] in (True, False)


@pytest.mark.parametrize("args,kernel_prefix", [
    (["--config", "4", "--kkt-n", "24", "--steps", "3", "--warmup", "1"], "csr_"),
    (["--config", "2", "--steps", "5", "--warmup", "1"], "csr_"),
])
def test_bench_lines_of_the_other_configs(args, kernel_prefix, tmp_path):
    """`bench.py --config 4` (generated nlpkkt160-shaped .mtx, here on a 24^3
    grid, through io_load_csr_cached) and `--config 2` (1M banded CSR,
    flushed) print the same kind of line as the headline."""
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(S.ROOT, "bench.py"),
                        "--no-cpu-baseline"] + args, capture_output=True,
                       text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["dtype"] == "f64"
    assert j["config"]["kernel"].startswith(kernel_prefix)
    assert j["rows_checked"] == 258
    roof = j["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-3
    assert set(j["extras"]) == {"csr_wave_row", "csr_subwave_row", "csr_stream"}
    if "--kkt-n" in args:
        assert "24^3 grid" in j["config"]["workload"]
        assert j["config"]["loaded_from"].startswith(".mtx text")
        assert j["config"]["nnz"] == 734664  # gen_kkt_mtx 24: after mirroring
