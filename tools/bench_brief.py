#!/usr/bin/env python3
"""Print the fields of a bench.py JSON line one looks at first (value, step, roofline, device time, parity)."""
import json
import sys


def main(path):
    line = [l for l in open(path) if l.startswith("{")]
    if not line:
        print("no JSON line in", path)
        return
    j = json.loads(line[-1])
    r = j.get("roofline", {})
    keys = ("value", "ms_per_step", "n_gpus")
    print(" ".join("%s=%s" % (k, j.get(k)) for k in keys if k in j),
          "| roofline %s frac=%s achieved=%s" % (r.get("kernel", ""), r.get("frac"), r.get("achieved")),
          "| parity", {k: v for k, v in j.get("parity", {}).items() if k != "by_decade"})
    if "kernel_time_share" in j:
        print("  kernel_time_share", {k: round(v, 3) for k, v in j["kernel_time_share"].items()})


if __name__ == "__main__":
    main(sys.argv[1])
