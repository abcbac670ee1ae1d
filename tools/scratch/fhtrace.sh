#!/bin/bash
# kernel + copy timeline of the bed hand-off (scratch)
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/fhtrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -o k -- python3 tools/scratch/fhbed.py reg > $OUT/log.txt 2>&1
find $OUT/t -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
find $OUT/t -name '*memory_copy_trace.csv' -exec cp {} $OUT/memory_copy_trace.csv \;
rm -rf $OUT/t
tail -2 $OUT/log.txt
python3 - <<'P'
import csv, os
out = os.path.join(os.getcwd(), "gpurun_out", "fhtrace")
ev = []
for r in csv.DictReader(open(out + "/kernel_trace.csv")):
    n = r["Kernel_Name"]
    tag = None
    for key, t in (("consolidate_count", "CNT"), ("consolidate_fill", "FIL"), ("consolidate_write", "WRT"), ("gene_suffstat_hc", "K2"),
                   ("gene_pvalue", "PV"), ("gene_assemble", "AS")):
        if key in n: tag = t
    if tag: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag))
for r in csv.DictReader(open(out + "/memory_copy_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "CP:" + r.get("Direction", "?")[:12]))
ev.sort()
t_end = ev[-1][1]
win = [e for e in ev if e[0] > t_end - 30_000_000 and e[0] < t_end - 26_000_000]
t0 = win[0][0]
for s, e, tag in win:
    print("%9.1f %9.1f %7.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, tag))
P
