"""The optional legs of a bench line cannot cost the line (VERDICT r05 next
#1): benchlib/legs.py with two gloo ranks on CPU -- no GPU, no product
library.  The GPU rehearsal of the real bench.py under the same injections is
tests/test_gpu_dist2_rehearsal.py."""
import json
import os
import signal
import socket
import subprocess
import sys
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "_legs_worker.py")


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _start(scenario, world=2, **env_extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE",
                        "SPMV_BENCH_INJECT")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               WORLD_SIZE=str(world), **env_extra)
    return [subprocess.Popen([sys.executable, WORKER, scenario],
                             env=dict(env, RANK=str(r)),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True) for r in range(world)]


def _finish(procs, timeout=120):
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    return outs


def _lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def test_healthy_run_prints_the_line_twice_and_the_last_one_is_complete():
    outs = _finish(_start("healthy"))
    assert [rc for rc, _, _ in outs] == [0, 0], outs
    lines = _lines(outs[0][1])
    assert len(lines) == 2 and _lines(outs[1][1]) == []  # rank 0 only
    first, last = lines
    assert first["provisional"] is True and first["value"] == 42.0
    assert first["legs_pending"] == ["first", "second", "third"]
    assert "provisional" not in last and last["legs_failed"] == []
    assert last["config"]["second"] == {"tag": "built", "sum": 3.0}
    assert last["config"]["released"] == 1  # cleanup ran
    assert set(last["legs_s"]) == {"first", "second", "third"}


@pytest.mark.parametrize("how", ["scenario", "inject"])
def test_a_local_failure_on_one_rank_makes_every_rank_skip_the_leg(how):
    """rank 1 alone fails in the LOCAL half of a leg: without the agreement
    rank 0 would wait in the leg's all-reduce for a peer that never comes;
    with it both skip, the next leg still runs, rc 0, legs_failed names the
    rank"""
    if how == "scenario":
        procs = _start("local_failure_on_rank1")
    else:
        procs = _start("healthy", SPMV_BENCH_INJECT="second:1:prepare")
    t0 = time.time()
    outs = _finish(procs)
    assert time.time() - t0 < 60
    assert [rc for rc, _, _ in outs] == [0, 0], outs
    last = _lines(outs[0][1])[-1]
    assert last["value"] == 42.0 and "provisional" not in last
    assert len(last["legs_failed"]) == 1, last
    assert "second" in last["legs_failed"][0]
    assert "rank(s) [1]" in last["legs_failed"][0]
    assert last["config"]["second"] is None
    assert last["config"]["third"] == {"tag": "third", "sum": 3.0}
    assert last["config"]["released"] == 1


def test_a_rank_that_hangs_inside_a_leg_is_ended_by_the_watchdog_with_the_line():
    """rank 1 never arrives at the leg's collective: rank 0's watchdog prints
    the final line (main measurement + the legs done so far + the deadline in
    legs_failed) and both ranks leave with exit code 0, within the leg's
    limit -- not the process group's timeout"""
    t0 = time.time()
    outs = _finish(_start("healthy", SPMV_BENCH_INJECT="second:1:hang",
                          LEGS_LIMIT="8"))
    assert time.time() - t0 < 60
    assert [rc for rc, _, _ in outs] == [0, 0], outs
    lines = _lines(outs[0][1])
    assert len(lines) == 2
    last = lines[-1]
    assert last["value"] == 42.0 and "provisional" not in last
    assert last["config"]["first"] == {"tag": "first", "sum": 3.0}
    assert any("second" in f and "limit" in f for f in last["legs_failed"]), last
    assert "second" in last["legs_s"]


def test_a_rank_that_dies_inside_a_leg_breaks_the_group_not_the_line():
    outs = _finish(_start("healthy", SPMV_BENCH_INJECT="second:1:die",
                          LEGS_LIMIT="20"))
    assert outs[1][0] == 13
    assert outs[0][0] == 0, outs[0]
    last = _lines(outs[0][1])[-1]
    assert last["value"] == 42.0 and "provisional" not in last
    assert any("second" in f for f in last["legs_failed"]), last
    # the group is broken: the next collective leg is not started
    assert last["config"]["third"] is None


def test_the_whole_run_has_a_deadline_too():
    outs = _finish(_start("healthy", SPMV_BENCH_INJECT="third:0:hang",
                          LEGS_LIMIT="60", LEGS_DEADLINE="12"))
    assert [rc for rc, _, _ in outs] == [0, 0], outs
    last = _lines(outs[0][1])[-1]
    assert any("deadline" in f or "limit" in f for f in last["legs_failed"])
    assert last["config"]["second"] == {"tag": "built", "sum": 3.0}


def test_killing_the_job_after_the_main_measurement_leaves_a_parsable_line():
    procs = _start("healthy", SPMV_BENCH_INJECT="first:0:hang,first:1:hang",
                   LEGS_LIMIT="300", LEGS_DEADLINE="300")
    try:
        line = ""
        while not line.startswith("{"):  # (gloo prints a banner first)
            line = procs[0].stdout.readline()  # the provisional line, flushed
            assert line, "rank 0 ended without a line"
        j = json.loads(line)
        assert j["provisional"] is True and j["value"] == 42.0
        assert j["legs_pending"] == ["first", "second", "third"]
    finally:
        for p in procs:
            p.send_signal(signal.SIGKILL)
        for p in procs:
            p.wait()


def test_a_run_stuck_before_its_line_exists_says_where_and_exits_3():
    """rule 5: rank 1 never joins the first collective of the MAIN
    measurement (what a broken fabric looks like on a node nobody has seen):
    instead of the process group's timeout and a traceback, rank 0 prints a
    failure record -- value null, the phase, seconds per phase -- and every
    rank exits with code 3"""
    t0 = time.time()
    outs = _finish(_start("main_stuck", LEGS_MAIN_LIMIT="6"))
    assert time.time() - t0 < 60
    assert [rc for rc, _, _ in outs] == [3, 3], outs
    rec = _lines(outs[0][1])[-1]
    assert rec["value"] is None and rec["failed"] is True and rec["n_gpus"] == 2
    assert rec["failed_in"].startswith("first step + result check")
    assert rec["phases_s"]["setup"] >= 0.4 and rec["metric"] == "toy"
    assert _lines(outs[1][1]) == []


def test_eight_ranks_agree_on_a_local_failure_of_one():
    """the world size of the driver's largest run: rank 5 alone fails in the
    local half of a leg; all eight skip it, the next leg runs on all eight"""
    outs = _finish(_start("healthy", world=8,
                          SPMV_BENCH_INJECT="second:5:prepare"), timeout=240)
    assert [rc for rc, _, _ in outs] == [0] * 8, [o[0] for o in outs]
    last = _lines(outs[0][1])[-1]
    assert last["config"]["first"] == {"tag": "first", "sum": 36.0}
    assert last["config"]["second"] is None
    assert last["config"]["third"] == {"tag": "third", "sum": 36.0}
    assert len(last["legs_failed"]) == 1
    assert "rank(s) [5]" in last["legs_failed"][0]
    assert all(_lines(o[1]) == [] for o in outs[1:])  # rank 0 alone prints
