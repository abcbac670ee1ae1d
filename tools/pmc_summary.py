#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (sum over dispatches) and print/write a table."""
import collections
import csv
import sys


def main(path, out=None, only="rvt"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if only and only not in k:
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
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
