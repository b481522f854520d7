#!/usr/bin/env bash
# All rocprofv3 evidence of a round in one gpurun call (run on the GPU box):
#   tools/profile_round.sh r04 [all|passes|rest]
# For each workload three passes of the same bench.py command (tools/profile.sh:
# kernel trace + stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE; counters never share
# a pass with the hip/hsa trace domains), summarised into
# profiles/<round>_<tag>.md + .traffic.json; plus the TCP->TCC request counters
# of the headline kernel (profiles/<round>_wn.l2req.json), the un-profiled
# bench lines and the 1-GPU run of the fixed 80M x 80M problem that N > 1
# lines use as their speed-up denominator.  Stops at the first failing step.
set -uo pipefail
round="${1:-r04}"
part="${2:-all}"   # all | passes | rest  (two gpurun calls when one is too long)
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$root"
mkdir -p gpurun_out profiles
# only what THIS call writes goes back (the snapshot also holds the round's
# committed files: copying those would put stale records over fresh ones)
touch gpurun_out/.round_start
keep() { mkdir -p gpurun_out/profiles_${round}; find profiles -maxdepth 1 -name "${round}_*" -newer gpurun_out/.round_start -exec cp -f {} gpurun_out/profiles_${round}/ \; ; }
step() { echo "== $*"; "$@" || { echo "FAILED: $*"; exit 1; }; }
prof() {  # prof <tag> <bench args...>
    local tag="$1"; shift
    step tools/profile.sh "${round}_$tag" --steps 20 "$@"
    step python3 tools/summarize_profile.py "gpurun_out/prof_${round}_$tag" \
        "profiles/${round}_$tag.md" > /dev/null
    echo "-- profiles/${round}_$tag.md written"
}
if [ "$part" != "rest" ]; then
prof wn_hll_tile_panels
prof w20 --window 1048576
prof w17 --window 131072
prof c2_banded1M_csr --config 2
prof c4_kkt_csr --config 4
prof banded10M_hll --family banded --kernel 1
# the reference's irregular classes (download-matrices.py: webbase / amazon /
# roadNet; dc1), autotuned pick of an HLL handle = the blocked copy
prof powerlaw4M --family powerlaw --rows-per-gpu 4000000 --nnz-row 3
prof hub1M --family hub --rows-per-gpu 1000000 --nnz-row 6 --window 4096
fi
if [ "$part" = "passes" ]; then
    keep
    echo "== passes done"; exit 0
fi
# counter passes run the layout of an UN-profiled selector run (--blocked-pin)
# -- the layout of the workload's trace pass when this call took it (`all`), so
# that every record of a workload describes ONE layout; else a fresh run's
pinned() {  # pinned <tag> <prof tag> <bench args...>: "--blocked-pin <pin>" or nothing
    local tag="$1" prof="gpurun_out/prof_${round}_$2/bench_kt.json"; shift 2
    local src="gpurun_out/${round}_pin_$tag.json"
    if [ -s "$prof" ]; then
        src="$prof"
    else
        python3 bench.py --no-extras --no-cpu-baseline --steps 5 "$@" \
            > "$src" 2> /dev/null || return 0
    fi
    local pin; pin=$(python3 tools/blocked_pin.py "$src")
    [ -n "$pin" ] && echo "--blocked-pin $pin"
}
PIN_WN=$(pinned wn wn_hll_tile_panels)
step tools/pmc.sh "${round}_l2req" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
    bench.py --no-extras --no-cpu-baseline --steps 20 $PIN_WN
step python3 tools/l2req_profile.py "gpurun_out/pmc_${round}_l2req" \
    "profiles/${round}_wn.l2req.json" > /dev/null
step tools/pmc.sh "${round}_tcc" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
    bench.py --no-extras --no-cpu-baseline --steps 20 $PIN_WN
# the same request counters for the power-law matrix (what bounds 22 %) ...
PL="--family powerlaw --rows-per-gpu 4000000 --nnz-row 3"
PIN_PL=$(pinned pl powerlaw4M $PL)
step tools/pmc.sh "${round}_pl_l2req" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
    bench.py --no-extras --no-cpu-baseline --steps 20 $PL $PIN_PL
step python3 tools/l2req_profile.py "gpurun_out/pmc_${round}_pl_l2req" \
    "profiles/${round}_powerlaw4M.l2req.json" > /dev/null
# ... and for one rank's shard of config 5 (10M x 80M; profiles/<round>_config5_shard.counters.json)
SH="tools/sweep.py --rows 10000000 --cols 80000000 --k 32 --windows 0 --hll-kernels 4 --csr-kernels= --iters 10"
step tools/pmc.sh "${round}_sh_l2req" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" $SH
step tools/pmc.sh "${round}_sh_tcc" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" $SH
step tools/pmc.sh "${round}_sh_fetch" "FETCH_SIZE" $SH
step python3 tools/shard_counters.py "$round" > /dev/null
# un-profiled lines of the round
python3 bench.py --strong --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline \
    > "profiles/${round}_strong_1gpu.json" 2> gpurun_out/${round}_strong.err || { echo "FAILED strong"; exit 1; }
python3 bench.py --config 2 --steps 30 > "profiles/${round}_bench_config2.json" 2> /dev/null || exit 1
python3 bench.py --config 4 --steps 20 > "profiles/${round}_bench_config4.json" 2> /dev/null || exit 1
python3 bench.py --steps 20 --warmup 5 > "profiles/${round}_bench_full.json" 2> /dev/null || exit 1
keep
echo "== done"
