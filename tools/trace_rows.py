#!/usr/bin/env python3
"""Chronological rows of a rocprofv3 run: kernel dispatches (kernel_trace.csv) and memory copies (memory_copy_trace.csv) merged,
with duration and the gap to the previous record.   usage: trace_rows.py <dir with the csv files> [--last N] [--grep substr]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    last = 80
    a = sys.argv[2:]
    grep = None
    while a:
        if a[0] == "--last":
            last = int(a[1])
        elif a[0] == "--grep":
            grep = a[1]
        a = a[2:]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rvt::", "")[:44]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + name))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
    rows.sort()
    if grep:
        idx = [i for i, r in enumerate(rows) if grep in r[2]]
        if idx:
            rows = rows[max(0, idx[-1] - last + 1): idx[-1] + 8]
    else:
        rows = rows[-last:]
    prev = None
    t0 = rows[0][0] if rows else 0
    for s, e, n in rows:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        print("%10.1f us  +%8.1f gap  %9.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, n))
        prev = e


if __name__ == "__main__":
    main()
