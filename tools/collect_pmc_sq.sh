#!/bin/bash
# Run on the GPU box (via gpurun): SQ counter summaries (matrix-pipe busy, VALU issue, wave cycles, instruction counts) of the
# four sufficient-statistics kernel families under their own bench workloads, one rocprofv3 --pmc pass each (kernel trace
# only).  -> gpurun_out/pmc_sq_r4/<workload>_sq.csv  (tools/pmc_summary.py tables; copied to profiles/r4_pmc_sq_*.csv)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_sq_r4
mkdir -p "$OUT"
export TMPDIR=/tmp
SET="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
run() {  # tag, bench args...
  local tag=$1; shift
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/$tag" -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-from-host "$@" > "$OUT/$tag.log" 2>&1
  F=$(find "$OUT/$tag" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$F" "$OUT/${tag}_sq.csv" > /dev/null
  rm -rf "$OUT/$tag"
  grep -c suffstat "$OUT/${tag}_sq.csv"
}
run hc                                   # gene_suffstat_hc (quantitative trait, hard calls: the headline)
run hcx --trait binary --samples 200000  # gene_suffstat_hcx (binary trait)
run lat --dosage                         # gene_suffstat_lat (lattice dosages)
run mfma --dosage --dosage-lattice 0     # gene_suffstat_mfma (dosages, lattice not stated: the general fp64 kernel)
RVT_HCX=0 run hcw --trait binary --samples 200000   # gene_suffstat_hcw (the one-wave weighted kernel, fallback)
ls -la "$OUT"
