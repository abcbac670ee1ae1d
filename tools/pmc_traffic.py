#!/usr/bin/env python3
"""Turn the two PMC summaries (tools/pmc_summary.py output for FETCH_SIZE and WRITE_SIZE, separate rocprofv3 passes
of `bench.py --steps S --warmup W`) into the HBM traffic figure bench.py reports as roofline.traffic.

FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE counts half of the bytes of wide
coalesced streaming reads (MI355X_MICROARCH.md, "HBM"), so it is doubled; WRITE_SIZE is taken as reported
(uncalibrated per the same guide).

usage: pmc_traffic.py <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <batches profiled> <workload key> <out.json> [kernel name]
(kernel name: the hard-call kernel the passes saw, e.g. gene_suffstat_hcx — bench.py only takes the figure for that kernel)
"""
import csv
import json
import sys


def load(path, col):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["kernel"]] = (int(r["dispatches"]), float(r[col]))
    return out


def main():
    fpath, wpath, batches, key, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    f, w = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
    fam = {}
    for k in f:
        if "gene_suffstat_lat" in k:
            tag = "suffstat_lat"
        elif "gene_suffstat_hc" in k:
            tag = "suffstat_hc"
        elif "gene_suffstat" in k:
            tag = "suffstat"
        else:
            tag = k.split("rvt::")[1].split("(")[0].split("<")[0]
        e = fam.setdefault(tag, {"dispatches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
        e["dispatches"] += f[k][0]
        e["fetch_bytes"] += 2.0 * 1024.0 * f[k][1]
        e["write_bytes"] += 1024.0 * w.get(k, (0, 0.0))[1]
    for e in fam.values():
        e["hbm_bytes_per_step"] = (e["fetch_bytes"] + e["write_bytes"]) / batches
        e["launches_per_step"] = e["dispatches"] / batches
    head = {"workload": key, "batches": batches}
    if len(sys.argv) > 6:
        head["kernel_name"] = sys.argv[6]
    json.dump({**head, "correction": "FETCH_SIZE x2 (gfx950), x1024 B; WRITE_SIZE x1024 B",
               "kernels": fam}, open(outp, "w"), indent=1)
    print(json.dumps({k: fam[k] for k in ("suffstat_hc", "suffstat_lat", "suffstat") if k in fam}))


if __name__ == "__main__":
    main()
