#!/bin/bash
B="python bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --missing-frac 0"
mkdir -p gpurun_out/r4b2
timeout 100 ./tools/hcx_bench check | tail -3
timeout 100 ./tools/hcx_bench spread | grep hcx
for t in 12 1 2 15; do
$B --tests $t > gpurun_out/r4b2/x_$t.json 2> gpurun_out/r4b2/x_$t.err
RVT_HCX=0 $B --tests $t > gpurun_out/r4b2/w_$t.json 2> gpurun_out/r4b2/w_$t.err
done
for f in x_12 w_12 x_1 w_1 x_2 w_2 x_15 w_15; do python - gpurun_out/r4b2/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d['roofline']['avg_launch_ms'], d.get('kernel_time_share',{}).get('device_ms_per_step'))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
