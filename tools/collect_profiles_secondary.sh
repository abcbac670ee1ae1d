#!/bin/bash
# Run on the GPU box (via gpurun): the secondary workloads of a round (family tests, meta tests, permutations, the 2-bit stream,
# the group stream, the decomposition) with their rocprofv3 kernel statistics.   -> gpurun_out/prof_<tag>_secondary/
set -u
TAG=${1:-r4}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_secondary
mkdir -p "$OUT"
export TMPDIR=/tmp
stats() {
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$name" -o k -- "$@" > "$OUT/${name}_result.txt" 2>&1
  find "$OUT/kt_$name" -name '*kernel_stats.csv' -exec cp {} "$OUT/${name}_kernel_stats.csv" \;
  rm -rf "$OUT/kt_$name"
}
stats famskat python3 tools/bench_famskat.py --samples 100000 --genes 128
stats famskat_dense python3 tools/bench_famskat.py --samples 100000 --genes 128 --dense
stats famskat_shuffled python3 tools/bench_famskat.py --samples 100000 --genes 128 --shuffle
stats metascore python3 tools/bench_metascore.py
stats metacov python3 tools/bench_metacov.py --reps 5
stats stream_bed python3 tools/bench_stream.py --bed --genes 512
python3 tools/bench_perm.py --genes 8 > "$OUT/perm_result.txt" 2>&1
python3 tools/bench_perm.py --genes 2 --alpha 1 --nperm 16384 >> "$OUT/perm_result.txt" 2>&1
python3 tools/bench_perm.py --genes 1 --exact >> "$OUT/perm_result.txt" 2>&1
python3 tools/bench_group_stream.py > "$OUT/group_stream.txt" 2>&1
python3 tools/bench_decompose.py --samples 3000 --kind grm > "$OUT/decompose.txt" 2>&1
python3 tools/bench_decompose.py --samples 100000 --kind family --install >> "$OUT/decompose.txt" 2>&1
./tools/rotgemm_bench bench > "$OUT/rotgemm.txt" 2>&1
ls -la "$OUT"
