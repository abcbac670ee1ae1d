#!/bin/bash
# Run on the GPU box (via gpurun): SQ counters of MetaCov's band product — the int8 kernel and the MXFP4 kernel of
# band_gemm.hip.h under tools/band_bench (4 096 heads, 3 000 markers, N = 500 000), one rocprofv3 --pmc pass (kernel trace
# only).  -> gpurun_out/prof_r6/pmc_sq_band.csv
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r6
mkdir -p "$OUT"
export TMPDIR=/tmp
SET="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/sqb" -o p -- ./tools/band_bench bench 500000 4096 3000 > "$OUT/pmc_sq_band.log" 2>&1
F=$(find "$OUT/sqb" -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py --kernels band_gemm "$F" "$OUT/pmc_sq_band.csv"
rm -rf "$OUT/sqb"
python3 - "$OUT/pmc_sq_band.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    busy = float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) / 32.0 / max(float(r["SQ_BUSY_CYCLES"]), 1.0)
    wc = max(float(r["SQ_WAVE_CYCLES"]), 1.0)
    print("%-60s dispatches %s  matrix pipe busy %.2f  VALU issuing %.2f  issue stalls %.2f  parked %.2f" % (
        r["kernel"][-60:], r["dispatches"], busy, float(r["SQ_ACTIVE_INST_VALU"]) / wc, float(r["SQ_WAIT_INST_ANY"]) / wc, float(r["SQ_WAIT_ANY"]) / wc))
PY
