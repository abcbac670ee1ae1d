#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals over a trailing window and a per-queue timeline.

usage: timeline.py <kernel_trace.csv> [--last-ms 150] [--rows 0]
"""
import csv
import sys
from collections import defaultdict


def short(name):
    if "gene_suffstat_hcx" in name:
        i = name.find("<")
        return "K2hcx" + name[i:name.find(">", i) + 1].replace(" ", "")
    if "gene_suffstat_hcw" in name:
        i = name.find("<")
        return "K2hcw" + name[i:name.find(",", i)].replace(" ", "") + ">"
    if "gene_suffstat_hc" in name:
        i = name.find("<")
        return "K2hc" + name[i:name.find(",", i)].replace(" ", "") + ">"
    for key, tag in (("gene_suffstat_panel", "K2p"), ("gene_suffstat_mfma", "K2"), ("gene_flags", "FL"),
                     ("burden_fallback", "BF"),
                     ("burden_collapse", "BU"), ("gene_assemble", "AS"), ("gene_tridiag", "TD"), ("gene_spectrum", "SP"),
                     ("gene_pvalue", "PV")):
        if key in name:
            if tag == "K2":
                i = name.find("<")
                return "K2" + name[i:name.find(">", i) + 1].replace(" ", "")
            return tag
    return None


def main():
    path = sys.argv[1]
    last_ms = 150.0
    rows = 0
    a = sys.argv[2:]
    while a:
        if a[0] == "--last-ms":
            last_ms = float(a[1])
        elif a[0] == "--rows":
            rows = int(a[1])
        a = a[2:]
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            tag = short(r["Kernel_Name"])
            if tag is None:
                continue
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag, r.get("Queue_Id", "?")))
    ev.sort()
    t_end = max(e[1] for e in ev)
    t0 = t_end - int(last_ms * 1e6)
    win = [e for e in ev if e[0] >= t0]
    tot = defaultdict(lambda: [0, 0.0])
    for s, e, tag, q in win:
        tot[tag][0] += 1
        tot[tag][1] += (e - s) / 1e6
    span = (max(e[1] for e in win) - min(e[0] for e in win)) / 1e6
    npv = tot["PV"][0] if "PV" in tot else 1
    print("window %.1f ms, %d batches (PV launches)" % (span, npv))
    print("%-22s %6s %10s %12s" % ("kernel", "calls", "total ms", "ms / batch"))
    for tag, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("%-22s %6d %10.2f %12.3f" % (tag, n, ms, ms / max(npv, 1)))
    for grp, pred in (("K2*", lambda t: t.startswith("K2")), ("stage2 (AS+TD+SP+PV)", lambda t: t in ("AS", "TD", "SP", "PV"))):
        # union of busy intervals
        iv = sorted((s, e) for s, e, t, q in win if pred(t))
        busy, cur_s, cur_e = 0, None, None
        for s, e in iv:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            busy += cur_e - cur_s
        print("%-22s busy %.2f ms of %.2f (%.0f%%)" % (grp, busy / 1e6, span, 100 * busy / 1e6 / span))
    if rows:
        base = win[0][0]
        for s, e, tag, q in win[-rows:]:
            print("%9.3f %9.3f q%-3s %-14s %8.3f" % ((s - base) / 1e6, (e - base) / 1e6, q, tag, (e - s) / 1e6))


if __name__ == "__main__":
    main()
