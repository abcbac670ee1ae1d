#!/bin/bash
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r4c1; mkdir -p $OUT
A="--steps 40 --warmup 8 --no-cpu-baseline --no-from-host --samples 50000 --m-lo 30 --m-hi 30 --genes 1024 --tests 1"
python3 bench.py $A > $OUT/bench.json 2> $OUT/bench.err
python3 - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['kernel_time_share'], d['roofline']['kernel'], d['roofline']['frac'])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o k -- python3 bench.py $A > $OUT/kt.log 2>&1
find $OUT/kt -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
TR=$(find $OUT/kt -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py "$TR" --last-ms 40 --rows 60 > $OUT/timeline.txt 2>&1
rm -rf $OUT/kt
cut -c1-150 $OUT/kernel_stats.csv | head -12
head -50 $OUT/timeline.txt
