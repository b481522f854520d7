#!/usr/bin/env bash
# Profile bench.py on the GPU box with rocprofv3 (run through gpurun).
#   tools/profile.sh <tag> [bench.py args...]
# Writes gpurun_out/prof_<tag>/{kt,fetch,write}/ (CSV).  Kernel trace + stats
# and each PMC counter are separate passes (MI355X_MICROARCH.md: TCC has 4
# slots, FETCH_SIZE takes 3, WRITE_SIZE 2; counters are never combined with
# the sys/hip trace domains).  The first pass (trace) lets the selector pick;
# the counter passes are pinned to that pick.
set -uo pipefail
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args=(--no-extras --no-cpu-baseline "$@")
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- \
    python3 "$root/bench.py" "${args[@]}" > "$out/bench_kt.json" 2> "$out/kt.err" || exit 1
# The counter passes run the layout the trace pass's selector settled on
# (config.blocked_pin): under the counters' serialised launches the selector
# can pick another candidate, and the passes would describe different kernels.
pin=$(python3 "$root/tools/blocked_pin.py" "$out/bench_kt.json")
[ -n "$pin" ] && args+=(--blocked-pin "$pin")
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- \
    python3 "$root/bench.py" "${args[@]}" > "$out/bench_fetch.json" 2> "$out/fetch.err" || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -- \
    python3 "$root/bench.py" "${args[@]}" > "$out/bench_write.json" 2> "$out/write.err" || exit 1
find "$out" -name "*.csv" | head -30
