#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (…_results.db) into the per-kernel stats table
(name, calls, total µs, average µs, %) that `rocprofv3 --kernel-trace --stats` reports."""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for r in rows:
            w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.4f" % r[4]])
    print("wrote", out_path, len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
