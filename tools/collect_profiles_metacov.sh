#!/bin/bash
# Run on the GPU box (via gpurun): the MetaCov band measured like the headline — the tool's JSON lines, the rocprofv3 kernel
# statistics of the same command and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) behind
# the HBM traffic per block call.   usage: tools/collect_profiles_metacov.sh <tag>  -> gpurun_out/prof_<tag>/metacov_*
set -u
TAG=${1:-r5}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 tools/bench_metacov.py > "$OUT/metacov_result.txt" 2> "$OUT/metacov.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_mc" -o k -- python3 tools/bench_metacov.py --window "" --window-dosage "" --no-cpu > "$OUT/metacov_kt.log" 2>&1
find "$OUT/kt_mc" -name '*kernel_stats.csv' -exec cp {} "$OUT/metacov_kernel_stats.csv" \;
rm -rf "$OUT/kt_mc"
REPS=3
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_mc_$C" -o p -- python3 tools/bench_metacov.py --window "" --window-dosage "" --no-cpu --reps $REPS > "$OUT/metacov_pmc_$C.log" 2>&1
  F=$(find "$OUT/pmc_mc_$C" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$F" "$OUT/metacov_pmc_$C.csv" > /dev/null
  rm -rf "$OUT/pmc_mc_$C"
done
# (REPS timed + 1 warm-up call per workload: REPS + 1 launches of every kernel of a block call)
python3 tools/pmc_traffic.py "$OUT/metacov_pmc_FETCH_SIZE.csv" "$OUT/metacov_pmc_WRITE_SIZE.csv" $((REPS + 1)) "metacov block N=500000,V=1024 (per batch: one fp64 call, one hard-call call on a block uploaded at once, one on a block filled column by column; cov_hc_prep_kernel<4,true> also counts the 1024 one-column passes behind the uploads)" "$OUT/pmc_traffic_metacov.json"
python3 - "$OUT/pmc_traffic_metacov.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
alg = 8.0 * 500000 * 1024
for k, e in j["kernels"].items():
    print("%-28s %6.2f GB per call (algorithmic block: %.2f GB) launches/call %.1f" % (k, e["hbm_bytes_per_step"] / 1e9, alg / 1e9, e["launches_per_step"]))
PY
ls -la "$OUT"
