#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (sum over dispatches) and print/write a table."""
import collections
import csv
import sys


def main(path, out=None, only="rvt", kernels=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if kernels:
            if not any(t in k for t in kernels):
                continue
            k = k.split("(")[0][-48:]
        elif only and only not in k:
            continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (k, row.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
    names = sorted({c for v in agg.values() for c in v})
    lines = [",".join(["kernel", "dispatches"] + names)]
    for k, v in agg.items():
        lines.append(",".join(['"%s"' % k, str(cnt[k])] + ["%.0f" % v.get(c, 0.0) for c in names]))
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    av = sys.argv[1:]
    kern = None
    if av and av[0] == "--kernels":  # comma-separated substrings of the kernel names to keep
        kern = av[1].split(",")
        av = av[2:]
    main(av[0], av[1] if len(av) > 1 else None, kernels=kern)
