#!/bin/bash
# Run on the GPU box (via gpurun): the round-4 evidence that is not a bench.py line of configs[3] (that one is
# tools/collect_profiles_config3.sh): the default bench line (configs[2]) with kernel statistics, configs[1], the isolated
# rates of gene_suffstat_hcx, the C++ host-feed measurement.   -> gpurun_out/prof_r4/
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r4
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
tail -c 300 "$OUT/bench_line.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o k -- python3 bench.py --no-cpu-baseline --no-from-host > "$OUT/kt.log" 2>&1
find "$OUT/kt" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/kt"
python3 bench.py --no-cpu-baseline --no-from-host --samples 50000 --m-lo 30 --m-hi 30 --genes 1024 --tests 1 > "$OUT/bench_config1.json" 2>> "$OUT/bench.err"
python3 bench.py --dosage --no-cpu-baseline --no-from-host > "$OUT/bench_dosage.json" 2>> "$OUT/bench.err"
python3 bench.py --dosage --dosage-float --no-cpu-baseline --no-from-host > "$OUT/bench_dosage_float.json" 2>> "$OUT/bench.err"
python3 bench.py --dosage --dosage-lattice 0 --no-cpu-baseline --no-from-host > "$OUT/bench_dosage_fp64.json" 2>> "$OUT/bench.err"
python3 bench.py --trait binary --no-cpu-baseline --no-from-host > "$OUT/bench_binary_n500k.json" 2>> "$OUT/bench.err"
{ echo "== tools/fdx_bench check"; ./tools/fdx_bench check | tail -12; echo "== tools/fdx_bench (N = 500 000)"; ./tools/fdx_bench; } > "$OUT/fdx_isolated.txt" 2>&1
{ echo "== tools/hcx_bench check"; ./tools/hcx_bench check | tail -12; echo "== tools/hcx_bench spread (the widths of a batch, N = 200 000)"; ./tools/hcx_bench spread; echo "== tools/hcx_bench spread miss (0.1 % of the entries imputed)"; ./tools/hcx_bench spread miss; } > "$OUT/hcx_isolated.txt" 2>&1
{ echo "== tools/host_feed_bench (N = 500 000, M = 50)"; ./tools/host_feed_bench; echo "== --registered"; ./tools/host_feed_bench --registered --modes bed,int8; } > "$OUT/host_feed_cpp.txt" 2>&1
ls -la "$OUT"
