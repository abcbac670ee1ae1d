#!/bin/bash
# Run on the GPU box (via gpurun): BASELINE configs[3] (binary trait, N = 200 000) — the bench lines (5 % / 0 % / 100 % of
# the genes with imputed columns), the rocprofv3 kernel statistics of the same command and the two PMC passes (FETCH_SIZE,
# WRITE_SIZE — separate runs, kernel trace only) that roofline.traffic comes from.
# usage: tools/collect_profiles_config3.sh <tag>  -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r4c3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
B="--trait binary --samples 200000"
python3 bench.py $B --steps 40 --warmup 8 > "$OUT/bench_config3.json" 2> "$OUT/bench.err"
tail -c 300 "$OUT/bench_config3.json"
python3 bench.py $B --steps 40 --warmup 8 --missing-frac 0 --no-cpu-baseline --no-from-host > "$OUT/bench_config3_missing_none.json" 2>> "$OUT/bench.err"
python3 bench.py $B --steps 40 --warmup 8 --missing-frac 1.0 --no-cpu-baseline --no-from-host > "$OUT/bench_config3_missing_all.json" 2>> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o k -- python3 bench.py $B --steps 40 --warmup 8 --no-cpu-baseline --no-from-host > "$OUT/kt.log" 2>&1
find "$OUT/kt" -name '*kernel_stats.csv' -exec cp {} "$OUT/config3_kernel_stats.csv" \;
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o p -- python3 bench.py $B --steps 3 --warmup 1 --no-cpu-baseline --no-from-host > "$OUT/pmc_$C.log" 2>&1
  F=$(find "$OUT/pmc_$C" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$F" "$OUT/pmc_config3_$C.csv" > /dev/null
done
python3 tools/pmc_traffic.py "$OUT/pmc_config3_FETCH_SIZE.csv" "$OUT/pmc_config3_WRITE_SIZE.csv" 4 "N=200000,genes=512,m=20..80,seed=20260002,tests=15,binary" "$OUT/pmc_traffic_config3.json" gene_suffstat_hcx
rm -rf "$OUT/kt" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
ls -la "$OUT"
