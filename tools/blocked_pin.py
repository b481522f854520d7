#!/usr/bin/env python3
"""print config.blocked_pin of the last JSON line of a bench.py output file
(empty when the line's kernel is not the 2-D blocked one): what
`bench.py --blocked-pin` takes to run exactly that layout again.

    pin=$(python3 tools/blocked_pin.py gpurun_out/prof_r04_w20/bench_kt.json)
"""
import json
import sys


def main():
    try:
        lines = [l for l in open(sys.argv[1]).read().splitlines()
                 if l.startswith("{")]
        print(json.loads(lines[-1])["config"].get("blocked_pin") or "")
    except (OSError, ValueError, KeyError, IndexError):
        print("")


if __name__ == "__main__":
    main()
