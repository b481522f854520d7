#!/usr/bin/env python3
"""After `tools/profile_round.sh <round> all` came back through gpurun: copy
the round's records from gpurun_out/profiles_<round>/ into profiles/, check
that every *.traffic.json / *.l2req.json names the tree's sources (kernel file
+ hip_common.h) and that its passes agree, and print the numbers the docs
quote.  Reports (autotune_*, selector_phases, irregular_kernel_stats) and the
N > 1 rehearsal lines are produced by their own tools and left alone.

    python tools/collect_round.py r04
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("_autotune", "_selector", "_bench_native", "_bench_force", "_irregular")


def blob(path):
    return subprocess.run(["git", "hash-object", path], capture_output=True,
                          text=True, cwd=ROOT).stdout.strip()


def main():
    rnd = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "profiles_" + rnd)
    for fn in sorted(os.listdir(src)):
        if not any(k in fn for k in KEEP):
            shutil.copy(os.path.join(src, fn), os.path.join(ROOT, "profiles", fn))
    common = blob("spmv_scpa_amd/csrc/hip_common.h")
    ok = True
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", rnd + "_*.traffic.json"))
                     + glob.glob(os.path.join(ROOT, "profiles", rnd + "_*.l2req.json"))):
        t = json.load(open(fn))
        ks = t["kernel_source"]
        good = (ks["blob"] == blob("spmv_scpa_amd/csrc/" + ks["file"])
                and ks.get("common_blob") == common
                and t.get("passes_agree", True))
        ok &= good
        print("%-40s %-30s %s %s" % (os.path.basename(fn), t["kernel"],
                                     "ok " if good else "STALE",
                                     t.get("blocked_schedule")))
    sys.path.insert(0, ROOT)
    from benchlib.common import last_json_line
    j = last_json_line(os.path.join(ROOT, "profiles", rnd + "_bench_full.json"))
    r = j["roofline"]
    print("headline: %.1f GFLOP/s, %.3f ms per step, frac %.4f, traffic %.3f GB"
          % (j["value"], j["ms_per_step"], r["frac"],
             (r["traffic"] or float("nan")) / 1e9))
    print("secondary:", r.get("secondary"))
    for k, v in j.get("extras", {}).items():
        print("  ", k, v)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
