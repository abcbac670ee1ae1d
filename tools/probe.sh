#!/bin/bash
# One parametrised measurement script for the GPU box (replaces the one-off tools/r4/*.sh).  Everything lands under
# gpurun_out/<tag>/; what is cited in DESIGN.md is copied to profiles/ by hand.
#
#   tools/probe.sh pvprof [samples] [genes]      cycle counters of gene_pvalue_kernel (profiling build, tools/pv_prof.py)
#   tools/probe.sh pvpmc  [samples] [genes]      SQ counters of the per-gene kernels on one isolated batch
#   tools/probe.sh bench  <tag> [bench.py args]  one bench.py line (env switches are inherited: RVT_PV_CUS=32 tools/probe.sh bench ...)
#   tools/probe.sh trace  <tag> [bench.py args]  the same under rocprofv3 --kernel-trace --stats
#   tools/probe.sh tests  <tag> [pytest args]    a subset of the GPU tests (env switches inherited)
set -u
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
cmd=${1:-}; shift || true
case "$cmd" in
pvprof)
  S=${1:-200000}; G=${2:-512}
  mkdir -p gpurun_out/pvprof
  RVT_LIBRARY=rvtests_amd/csrc/librvtests_amd_prof.so python3 tools/pv_prof.py --samples $S --genes $G 2>&1 | tee gpurun_out/pvprof/pvprof_${S}_${G}.txt
  ;;
pvpmc)
  S=${1:-200000}; G=${2:-512}
  OUT=$ROOT/gpurun_out/pvpmc; mkdir -p $OUT
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAVES"; do
    i=$((i+1))
    rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 tools/pv_prof.py --samples $S --genes $G > $OUT/p$i.log 2>&1
    F=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
    python3 tools/pmc_summary.py --kernels pvalue,spectrum,tridiag,assemble "$F" | tee -a $OUT/summary_${S}_${G}.txt
    rm -rf $OUT/p$i
  done
  ;;
bench)
  TAG=$1; shift
  mkdir -p gpurun_out/bench
  python3 bench.py "$@" > gpurun_out/bench/$TAG.json 2> gpurun_out/bench/$TAG.err
  tail -c 600 gpurun_out/bench/$TAG.err; python3 tools/bench_brief.py gpurun_out/bench/$TAG.json
  ;;
trace)
  TAG=$1; shift
  OUT=$ROOT/gpurun_out/trace_$TAG; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
  python3 tools/bench_brief.py $OUT/bench.json
  F=$(find $OUT -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp "$F" $OUT/kernel_stats.csv && head -14 $OUT/kernel_stats.csv | cut -c1-200
  find $OUT -name '*.csv' ! -name 'kernel_stats.csv' -size +2M -delete
  ;;
tests)
  TAG=$1; shift
  mkdir -p gpurun_out/tests
  python3 -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -15 | tee gpurun_out/tests/$TAG.txt
  ;;
*)
  sed -n 2,12p "$0"; exit 2;;
esac
