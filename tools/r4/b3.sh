#!/bin/bash
B="python bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --missing-frac 0"
mkdir -p gpurun_out/r4b3
for pv in 0 32 64 96 128; do
RVT_PV_CUS=$pv $B > gpurun_out/r4b3/x_$pv.json 2> gpurun_out/r4b3/x_$pv.err
RVT_PV_CUS=$pv RVT_HCX=0 $B > gpurun_out/r4b3/w_$pv.json 2> gpurun_out/r4b3/w_$pv.err
done
for f in x_0 w_0 x_32 w_32 x_64 w_64 x_96 w_96 x_128 w_128; do python - gpurun_out/r4b3/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d['roofline']['avg_launch_ms'], d.get('kernel_time_share',{}).get('device_ms_per_step'))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
