#!/bin/bash
# Run on the GPU box (via gpurun): the two PMC passes (FETCH_SIZE, WRITE_SIZE — separate runs, kernel trace only) of
# `bench.py --dosage`, summarised as profiles/r3_pmc_traffic_dosage.json's source -> gpurun_out/pmc_dosage/
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_dosage
mkdir -p "$OUT"
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o p -- python3 bench.py --dosage --steps 3 --warmup 1 --no-cpu-baseline --no-from-host > "$OUT/pmc_$C.log" 2>&1
  F=$(find "$OUT/pmc_$C" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$F" "$OUT/pmc_dosage_$C.csv" > /dev/null
  rm -rf "$OUT/pmc_$C"
done
python3 tools/pmc_traffic.py "$OUT/pmc_dosage_FETCH_SIZE.csv" "$OUT/pmc_dosage_WRITE_SIZE.csv" 4 "N=500000,genes=512,m=20..80,seed=20260002,tests=15,dosage,lattice=1000" "$OUT/pmc_traffic_dosage.json"
ls -la "$OUT"
