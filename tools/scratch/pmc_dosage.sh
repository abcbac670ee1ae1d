#!/bin/bash
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/pmc_dosage
mkdir -p $OUT
python3 bench.py --dosage --steps 20 --warmup 5 --no-cpu-baseline --no-from-host > $OUT/bench_dosage.json 2> $OUT/bench.err
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p_$T" -o p -- python3 bench.py --dosage --steps 2 --warmup 1 --no-cpu-baseline --no-from-host > "$OUT/$T.log" 2>&1
  F=$(find "$OUT/p_$T" -name '*counter_collection.csv' | head -1)
  python3 - "$F" <<'PY' > "$OUT/$T.txt"
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:12]:
    print(k, n[k], dict(acc[k]))
PY
  rm -rf "$OUT/p_$T"
done
