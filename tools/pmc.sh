#!/usr/bin/env bash
# PMC counters for an arbitrary python command on the GPU box:
#   tools/pmc.sh <tag> "<counters space separated>" <python args...>
set -uo pipefail
tag="$1"; ctr="$2"; shift 2
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/pmc_$tag"
mkdir -p "$out"
case "$1" in /*) ;; *) set -- "$root/$1" "${@:2}" ;; esac  # script path: absolute
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out" -- python3 "$@" > "$out/stdout.txt" 2> "$out/stderr.txt"
echo "rc=$?"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
d = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    d[(r["Kernel_Name"].split("(")[0][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
summary = {}
for k, v in sorted(d.items()):
    if len(v) >= 3:
        print("%-52s %-16s n=%3d avg=%.4g" % (k[0], k[1], len(v), sum(v) / len(v)))
        # bench.py's own loop comes last: average the last 20 launches only
        tail = v[-20:]
        summary.setdefault(k[0], {})[k[1]] = {"n": len(v), "avg": sum(v) / len(v),
                                              "avg_last20": sum(tail) / len(tail)}
import json, os
line = None
try:
    for l in open(sys.argv[1] + "/stdout.txt"):
        if l.startswith("{"):
            line = json.loads(l)
except (OSError, ValueError):
    pass
json.dump({"counters": summary, "bench": line},
          open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY
