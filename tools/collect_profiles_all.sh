#!/bin/bash
# Run on the GPU box (via gpurun): everything committed under profiles/ for one round.
# usage: tools/collect_profiles_all.sh <tag>    -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
tools/collect_profiles.sh "$TAG" > "$OUT/collect.log" 2>&1
stats() {  # stats <name> <program args...>: rocprofv3 kernel statistics of one tool run
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$name" -o k -- "$@" > "$OUT/$name.log" 2>&1
  find "$OUT/kt_$name" -name '*kernel_stats.csv' -exec cp {} "$OUT/${name}_kernel_stats.csv" \;
  rm -rf "$OUT/kt_$name"
}
stats famskat python3 tools/bench_famskat.py --samples 100000 --genes 128
stats famskat_dense python3 tools/bench_famskat.py --samples 100000 --genes 128 --dense
stats famskat_shuffled python3 tools/bench_famskat.py --samples 100000 --genes 128 --shuffle
stats metascore python3 tools/bench_metascore.py
tools/collect_profiles_metacov.sh "$TAG" > "$OUT/collect_metacov.log" 2>&1
stats perm python3 tools/bench_perm.py --genes 4
stats stream_bed python3 tools/bench_stream.py --bed --genes 512
python3 tools/bench_decompose.py --samples 3000 --kind grm > "$OUT/decompose.txt" 2>&1
python3 tools/bench_decompose.py --samples 12000 --kind family >> "$OUT/decompose.txt" 2>&1
python3 tools/bench_decompose.py --samples 100000 --kind family --install >> "$OUT/decompose.txt" 2>&1
# a dense GRM through the tridiagonal form (phase times on stderr), with the numpy CPU baseline
RVT_TRIDIAG_TRACE=1 python3 tools/bench_decompose.py --samples 12000 --kind grm 2>&1 | grep "^{\|^\[rvt\]" > "$OUT/decompose_dense.txt"
RVT_TRIDIAG_TRACE=1 python3 tools/bench_decompose.py --samples 24000 --kind grm 2>&1 | grep "^{\|^\[rvt\]" >> "$OUT/decompose_dense.txt"
for t in k2hc_bench k2hcw_bench k2lat_bench rotgemm_bench gemm64_bench; do  # micro-benchmarks: built here when the snapshot has no binary
  [ -x tools/$t ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/$t.hip -o tools/$t > "$OUT/build_$t.log" 2>&1
done
./tools/k2hc_bench bench > "$OUT/k2hc_isolated.txt" 2>&1
./tools/k2hcw_bench > "$OUT/k2hcw_isolated.txt" 2>&1
./tools/k2hcw_bench spread >> "$OUT/k2hcw_isolated.txt" 2>&1
./tools/k2lat_bench > "$OUT/k2lat_isolated.txt" 2>&1
./tools/k2lat_bench spread >> "$OUT/k2lat_isolated.txt" 2>&1
./tools/rotgemm_bench bench > "$OUT/rotgemm.txt" 2>&1
./tools/gemm64_bench check > "$OUT/gemm64.txt" 2>&1
./tools/gemm64_bench bench 500000 1024 >> "$OUT/gemm64.txt" 2>&1
./tools/gemm64_bench bench 500000 512 >> "$OUT/gemm64.txt" 2>&1
./tools/gemm64_bench bench 200000 2048 >> "$OUT/gemm64.txt" 2>&1
# the per-gene stages on one isolated batch: cycle counters of the profiling build and SQ counters (tools/probe.sh)
tools/probe.sh pvprof 200000 512 > "$OUT/pvprof.txt" 2>&1
tools/probe.sh pvpmc 200000 512 > "$OUT/pmc_sq_pvalue.csv" 2>&1
python3 tools/bench_perm.py --genes 8 > "$OUT/perm_result.txt" 2>&1
python3 tools/bench_perm.py --genes 2 --alpha 1 --nperm 16384 >> "$OUT/perm_result.txt" 2>&1
python3 tools/bench_perm.py --genes 1 --exact >> "$OUT/perm_result.txt" 2>&1
python3 tools/bench_group_stream.py > "$OUT/group_stream.txt" 2>&1
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --missing-frac 1.0 > "$OUT/bench_missing_all.json" 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --samples 50000 --m-lo 30 --m-hi 30 --genes 1024 --tests 1 > "$OUT/bench_config1.json" 2>/dev/null
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-from-host --samples 50000 --m-lo 30 --m-hi 30 --genes 1024 --tests 1 > "$OUT/bench_config1_100steps.json" 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --trait binary --samples 200000 > "$OUT/bench_config3_binary.json" 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --no-from-host --dosage > "$OUT/bench_dosage.json" 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --dosage --dosage-lattice 0 > "$OUT/bench_dosage_fp64.json" 2>/dev/null
stats dosage python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --dosage
ls -la "$OUT"
