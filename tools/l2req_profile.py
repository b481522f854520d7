#!/usr/bin/env python3
"""profiles/<name>.l2req.json from a tools/pmc.sh pass of bench.py with
TCP_TCC_READ_REQ_sum (+ latency): the L2 line requests per launch of the
headline kernel, keyed like the traffic profiles (workload, kernel, source
blob) so that bench.py's secondary roofline cannot quote another build.

    tools/pmc.sh l2req "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \\
        bench.py --no-extras --no-cpu-baseline --steps 20
    python tools/l2req_profile.py gpurun_out/pmc_l2req profiles/r03_wn.l2req.json
"""
import json
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    s = json.load(open(src + "/summary.json"))
    cfg = s["bench"]["config"]
    # the device function of bench.py's own loop: the one of the schedule the
    # line reports (the selector also ran the other schedules before it)
    func = {"sweep": "k_tiles_sweep", "chain": "k_tiles_chain",
            "steps": "k_tiles_step"}[cfg["blocked_schedule"]]
    best = None
    for kname, ctr in s["counters"].items():
        if func in kname and "TCP_TCC_READ_REQ_sum" in ctr:
            if best is None or ctr["TCP_TCC_READ_REQ_sum"]["n"] > best[1]["TCP_TCC_READ_REQ_sum"]["n"]:
                best = (kname, ctr)
    if not best:
        raise SystemExit("no blocked-path kernel with TCP_TCC_READ_REQ_sum in "
                         + src)
    kname, ctr = best
    req = ctr["TCP_TCC_READ_REQ_sum"]["avg_last20"]
    lat = ctr.get("TCP_TCC_READ_REQ_LATENCY_sum", {}).get("avg_last20")
    out = {"workload": cfg["workload"], "bench_kernel": cfg["kernel"],
           "kernel": kname, "kernel_source": cfg.get("kernel_source"),
           "blocked_layout": cfg.get("blocked_layout"),
           "blocked_schedule": cfg.get("blocked_schedule"),
           "kernel_choice": cfg.get("kernel_choice"),
           "requests_per_launch": req,
           "mean_latency_cycles": (lat / req) if lat else None,
           "launches": ctr["TCP_TCC_READ_REQ_sum"]["n"],
           "nnz": cfg.get("nnz_global"),
           "kernel_ms_avg_profiled": s["bench"]["roofline"]["kernel_ms_avg"],
           "source": dst.split("/")[-1],
           "counters": "rocprofv3 --pmc TCP_TCC_READ_REQ_sum "
                       "TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace "
                       "(tools/pmc.sh), mean of the last 20 launches"}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
