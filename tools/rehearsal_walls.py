#!/usr/bin/env python3
"""profiles/<round>_rehearsal_walls.md from the rehearsal runs of a round:

    tools/rehearsal_walls.py r06 gpurun_out/r6c15 > profiles/r06_rehearsal_walls.md

The directory holds `<tag>.json` (stdout of the bench command: provisional
line + final line) and `<tag>.wall` ("rc=0 wall=56 s") for the tags below."""
import json
import os
import sys

RUNS = (
    ("reh4_torchrun", "python -m torch.distributed.run --nnodes=1 "
     "--nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py "
     "--gpus 4 --steps 20 --warmup 5 --backend gloo"),
    ("reh2_own", "python bench.py --gpus 2 --steps 20 --warmup 5 "
     "--backend gloo"),
)


def main():
    rnd, src = sys.argv[1], sys.argv[2]
    out = ["# Rehearsals of the N > 1 bench command on the 1-GPU box (%s)" % rnd,
           "",
           "`--backend gloo`: the ranks share the one card and the y fragments "
           "travel through host memory, so",
           "the TIMINGS of the measurement mean nothing; what these records "
           "show is the control flow at FULL",
           "size (10M rows per rank, columns anywhere) and how long the legs "
           "take on the driver's clock.",
           "At most 6 processes may touch the card on a test box, so 4 ranks "
           "is the largest full rehearsal (a 6-rank",
           "torchrun attempt was ended by the box's process guard: 7 "
           "processes had the GPU open);",
           "the 8-rank control flow is covered by the gloo tests on CPU "
           "(tests/test_dist_gloo.py, world 8) and",
           "by 8 logical devices in one process (tests/test_gpu_mgpu.py, "
           "config 5 whole).", "",
           "| command | rc | wall (s) | lines printed | legs_s | legs_failed / "
           "skipped |", "|---|---|---|---|---|---|"]
    notes = []
    for tag, cmd in RUNS:
        path = os.path.join(src, tag + ".json")
        if not os.path.exists(path):
            continue
        lines = [json.loads(l) for l in open(path).read().splitlines()
                 if l.startswith("{")]
        wall = open(os.path.join(src, tag + ".wall")).read().split()
        j, c = lines[-1], lines[-1]["config"]
        out.append("| `%s` | %s | %s | %d (first provisional: %s; legs_pending "
                   "%s) | %s | %s / %s |" % (
                       cmd, wall[0].split("=")[1], wall[1].split("=")[1],
                       len(lines), lines[0].get("provisional"),
                       lines[0].get("legs_pending"), json.dumps(j["legs_s"]),
                       j["legs_failed"], j["legs_skipped"]))
        fam = c.get("family_variants") or {}
        notes += ["", "`%s`: value %.2f (%s); phases_s %s; arrangements: %s; "
                  "value_best %s; exchange_ms_alone %s; strong: %s; "
                  "family_variants: %s" % (
                      tag, j["value"], c["exchange_arrangement"],
                      json.dumps(j.get("phases_s")),
                      json.dumps(c["arrangements"]), j.get("value_best"),
                      c["exchange_ms_alone"], json.dumps(c["strong"]),
                      json.dumps({k: [v["kernel"], v["ms_per_step"]]
                                  for k, v in fam.items()}))]
    print("\n".join(out + notes))


if __name__ == "__main__":
    main()
