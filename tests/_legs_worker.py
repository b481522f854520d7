"""One rank of tests/test_bench_legs.py: the optional-leg machinery of
bench.py (benchlib/legs.py) with a toy job -- gloo on CPU, no GPU, no product
library.  argv: scenario name.  Rank 0 prints a provisional line, runs three
legs and prints the final line through the LegRunner."""
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from benchlib.legs import LegRunner  # noqa: E402


def main():
    scenario = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
    out = {"value": 42.0, "config": {}}
    legs = LegRunner(rank, world, time.time(), dist, None, True,
                     budget_s=float(os.environ.get("LEGS_BUDGET", "60")),
                     deadline_s=float(os.environ.get("LEGS_DEADLINE", "30")))

    def emit(provisional=False):
        with legs.lock:
            line = dict(out, legs_failed=list(legs.failed),
                        legs_skipped=list(legs.skipped),
                        legs_s=dict(legs.seconds))
            if provisional:
                line.update(provisional=True, legs_pending=list(legs.pending))
        print(json.dumps(line), flush=True)

    if scenario == "main_stuck":
        # the line does not exist yet: rank 1 never joins the first collective
        legs.main_limit_s = float(os.environ.get("LEGS_MAIN_LIMIT", "6"))
        legs.failure_record = lambda: {"metric": "toy"}
        legs.start_watchdog()
        legs.phase("setup")
        time.sleep(0.5)
        legs.phase("first step + result check (first exchange of y)")
        if rank == 1:
            time.sleep(600)
        t = torch.tensor([1.0])
        dist.all_reduce(t)
        raise SystemExit("not reached")
    legs.emit_final = emit
    legs.announce(["first", "second", "third"])
    if rank == 0:
        emit(provisional=True)
    legs.start_watchdog()

    def allreduce(tag):
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        return {"tag": tag, "sum": float(t.item())}

    def local_build():
        if scenario == "local_failure_on_rank1" and rank == 1:
            raise MemoryError("rank 1 cannot build its alternative")
        return "built"

    released = []
    out["config"]["first"] = legs.run("first", lambda: allreduce("first"))
    out["config"]["second"] = legs.run(
        "second", lambda b: allreduce(b), prepare=[local_build],
        cleanup=lambda *b: released.append(b), limit_s=float(
            os.environ.get("LEGS_LIMIT", "8")))
    out["config"]["released"] = len(released)
    out["config"]["third"] = legs.run("third", lambda: allreduce("third"))
    if not legs.broken:
        dist.destroy_process_group()
    legs.finish()
    if legs.broken:
        os._exit(0)


if __name__ == "__main__":
    main()
