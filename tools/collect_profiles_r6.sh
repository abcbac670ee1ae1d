#!/bin/bash
# Run on the GPU box (via gpurun): what round 6 commits under profiles/ (r6_*).
# usage: tools/collect_profiles_r6.sh   -> gpurun_out/prof_r6/
set -u
TAG=r6
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
tools/collect_profiles.sh "$TAG" > "$OUT/collect.log" 2>&1                      # headline: line, kernel statistics, PMC traffic
tools/collect_profiles_metacov.sh "$TAG" > "$OUT/collect_metacov.log" 2>&1      # MetaCov blocks (+ the window lines of the tool)
tools/collect_profiles_metacov_window.sh "$TAG" 200,1000,3000 > "$OUT/collect_metacov_window.log" 2>&1
tools/collect_profiles_config3.sh "$TAG" > "$OUT/collect_config3.log" 2>&1
stats() {
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$name" -o k -- "$@" > "$OUT/$name.log" 2>&1
  find "$OUT/kt_$name" -name '*kernel_stats.csv' -exec cp {} "$OUT/${name}_kernel_stats.csv" \;
  rm -rf "$OUT/kt_$name"
}
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --samples 50000 --m-lo 30 --m-hi 30 --genes 1024 --tests 1 > "$OUT/bench_config1.json" 2>/dev/null
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-from-host --samples 50000 --m-lo 30 --m-hi 30 --genes 1024 --tests 1 > "$OUT/bench_config1_100steps.json" 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --missing-frac 1.0 > "$OUT/bench_missing_all.json" 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --dosage > "$OUT/bench_dosage.json" 2>/dev/null
python3 tools/bench_famskat.py --samples 100000 --genes 128 > "$OUT/famskat_result.txt" 2>&1
python3 tools/bench_metascore.py > "$OUT/metascore_result.txt" 2>&1
RVT_TRIDIAG_TRACE=1 python3 tools/bench_decompose.py --samples 12000 --kind grm 2>&1 | grep "^{\|^\[rvt\]" > "$OUT/decompose_dense.txt"
RVT_TRIDIAG_TRACE=1 python3 tools/bench_decompose.py --samples 24000 --kind grm 2>&1 | grep "^{\|^\[rvt\]" >> "$OUT/decompose_dense.txt"
./tools/band_bench check > "$OUT/band_bench.txt" 2>&1
./tools/band_bench bench 500000 1024 1000 >> "$OUT/band_bench.txt" 2>&1
./tools/band_bench bench 500000 4096 3000 >> "$OUT/band_bench.txt" 2>&1
./tools/gemm64_bench check > "$OUT/gemm64.txt" 2>&1
./tools/gemm64_bench bench 500000 1024 >> "$OUT/gemm64.txt" 2>&1
(rvtests_amd/csrc/host/host_driver --synthetic-meta 500000 8000 200; rvtests_amd/csrc/host/host_driver --synthetic-meta 500000 12000 1000; rvtests_amd/csrc/host/host_driver --synthetic-meta 500000 30000 20) > "$OUT/metacov_dropin.txt" 2>&1
tools/kernel_units.sh > "$OUT/kernel_units.txt" 2>&1
ls -la "$OUT"
