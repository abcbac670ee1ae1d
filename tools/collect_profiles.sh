#!/bin/bash
# Run on the GPU box (via gpurun): the bench line, the rocprofv3 kernel statistics of the same command and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE — separate runs, kernel-trace only) that the roofline's `traffic` comes from.
# usage: tools/collect_profiles.sh <tag>        -> gpurun_out/prof_<tag>/{bench_line.json,kernel_stats.csv,pmc_*.csv}
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench_line.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o k -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host > "$OUT/kt.log" 2>&1
cp "$OUT"/kt/k_kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || find "$OUT/kt" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-from-host > "$OUT/pmc_$C.log" 2>&1
  F=$(find "$OUT/pmc_$C" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$F" "$OUT/pmc_$C.csv" > /dev/null
done
python3 tools/pmc_traffic.py "$OUT/pmc_FETCH_SIZE.csv" "$OUT/pmc_WRITE_SIZE.csv" 4 "N=500000,genes=512,m=20..80,seed=20260002,tests=15" "$OUT/pmc_traffic.json"
rm -rf "$OUT/kt" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
ls -la "$OUT"
