#!/usr/bin/env bash
# Run GPU steps one after another on the box (through gpurun); stop at the
# first step that was killed or timed out (no further GPU work after a kill),
# carry on after an ordinary non-zero exit (a failing test).
#   tools/gpu_steps.sh <tag> "<cmd1>" "<cmd2>" ...
# Each step's stdout/stderr go to gpurun_out/<tag>_<k>.log.
set -u
tag="$1"; shift
mkdir -p gpurun_out
k=0
for cmd in "$@"; do
    k=$((k + 1))
    log="gpurun_out/${tag}_${k}.log"
    echo "== step $k: $cmd" | tee "$log"
    timeout -k 10 "${STEP_TIMEOUT:-600}" bash -c "$cmd" >> "$log" 2>&1
    rc=$?
    echo "== step $k rc=$rc" | tee -a "$log"
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then
        echo "step $k was killed (rc=$rc): stopping" | tee -a "$log"
        exit $rc
    fi
done
exit 0
