#!/bin/bash
# kernel + copy timeline of the VCF text hand-off (scratch): fhtrace2.sh [bgen]
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/fhtrace2
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -o k -- python3 tools/scratch/fhvcf.py $1 > $OUT/log.txt 2>&1
find $OUT/t -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
find $OUT/t -name '*memory_copy_trace.csv' -exec cp {} $OUT/memory_copy_trace.csv \;
rm -rf $OUT/t
tail -2 $OUT/log.txt
python3 - <<'P'
import csv, os
out = os.path.join(os.getcwd(), "gpurun_out", "fhtrace2")
ev = []
for r in csv.DictReader(open(out + "/kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "rvt::" not in n: continue
    tag = n.split("rvt::")[1].split("(")[0].split("<")[0][:28]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag))
cp = []
for r in csv.DictReader(open(out + "/memory_copy_trace.csv")):
    cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
cp.sort()
# merge consecutive copies (one per record) into bursts
bursts = []
for s, e in cp:
    if bursts and s - bursts[-1][1] < 20000: bursts[-1][1] = e; bursts[-1][2] += 1
    else: bursts.append([s, e, 1])
for s, e, n in bursts: ev.append((s, e, "COPY x%d" % n))
ev.sort()
t_end = ev[-1][1]
win = [e for e in ev if e[0] > t_end - 40_000_000 and e[0] < t_end - 28_000_000 and (e[1] - e[0]) > 30000]
t0 = win[0][0]
for s, e, tag in win:
    print("%9.1f %9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, tag))
P
