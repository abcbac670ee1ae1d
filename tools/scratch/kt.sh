#!/bin/bash
# kernel statistics of bench runs (scratch)
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/kt_now
mkdir -p $OUT
run() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$name" -o k -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host "$@" > "$OUT/$name.log" 2>&1
  find "$OUT/kt_$name" -name '*kernel_stats.csv' -exec cp {} "$OUT/${name}_kernel_stats.csv" \;
  rm -rf "$OUT/kt_$name"
  tail -1 "$OUT/$name.log" | cut -c1-300
}
run small --samples 8000
run small_binary --trait binary --samples 8000
run binary --trait binary --samples 200000
