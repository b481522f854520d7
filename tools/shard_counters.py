#!/usr/bin/env python3
"""profiles/<round>_config5_shard.counters.json from the three tools/pmc.sh
passes of tools/profile_round.sh over one rank's shard of config 5 (10M rows x
80M columns, blocked copy): L2 line requests and their latency, L2 hits /
misses, L2 fill bytes -- and what follows from them (the slot model's time, the
miss rate, fill bytes against the algorithmic ones).

    python tools/shard_counters.py r04
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd = sys.argv[1]
    passes = {t: json.load(open(os.path.join(
        ROOT, "gpurun_out", "pmc_%s_sh_%s" % (rnd, t), "summary.json")))["counters"]
        for t in ("l2req", "tcc", "fetch")}
    # the blocked kernel the selector settled on: the k_tiles_* function with
    # the most launches in the run
    tiles = {k: v for k, v in passes["l2req"].items() if "k_tiles_" in k}
    kernel = max(tiles, key=lambda k: tiles[k]["TCP_TCC_READ_REQ_sum"]["n"])
    launches = tiles[kernel]["TCP_TCC_READ_REQ_sum"]["n"]
    c = {}
    for v in passes.values():
        for name, x in v[kernel].items():
            c[name] = x["avg"]
    req, lat = c["TCP_TCC_READ_REQ_sum"], c["TCP_TCC_READ_REQ_LATENCY_sum"]
    M, N, K = 10_000_000, 80_000_000, 32
    out = {
        "workload": "one rank's shard of config 5: random HLL %d x %d, %d "
                    "nnz/row, columns anywhere, seed 42 (tools/sweep.py --rows "
                    "%d --cols %d --k %d --windows 0 --hll-kernels 4)"
                    % (M, N, K, M, N, K),
        "kernel": kernel, "launches": launches,
        "how": "tools/pmc.sh: one rocprofv3 --pmc pass per counter group "
               "(with --kernel-trace only), mean over the run's launches",
        "counters": c,
        "derived": {
            "mean_latency_cycles": round(lat / req, 1),
            "slot_model_ms": round(lat / (256 * 107) / 2.4e9 * 1e3, 3),
            "slot_model": "latency sum / (256 CUs x 107 outstanding line "
                          "requests) / 2.4 GHz",
            "l2_miss_rate": round(c["TCC_MISS_sum"] / c["TCC_REQ_sum"], 3),
            "l2_fill_bytes": c["FETCH_SIZE"] * 2 * 1024,
            "algorithmic_bytes": 12 * M * K + 12 * (M // 32) + 8 * M + 8 * N,
            "x_line_reuse_per_phase":
                "rows in LDS per XCD (32 CUs x ~19 600) x 32 entries x 16 "
                "doubles per 128-B line / 8e7 columns = 4.0: a quarter of the "
                "gathers miss the L2 whatever the panel width or the "
                "XCD-to-column assignment; the other 7 XCDs find the line in "
                "the Infinity Cache (all XCDs sweep the panels in one order)",
        },
    }
    dst = os.path.join(ROOT, "profiles", "%s_config5_shard.counters.json" % rnd)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["derived"], indent=1))


if __name__ == "__main__":
    main()
