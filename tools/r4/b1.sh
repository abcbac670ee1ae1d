#!/bin/bash
B="python bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host"
mkdir -p gpurun_out/r4b1
$B > gpurun_out/r4b1/c3.json 2> gpurun_out/r4b1/c3.err
$B --missing-frac 0 > gpurun_out/r4b1/c3_m0.json 2> gpurun_out/r4b1/c3_m0.err
$B --missing-frac 1.0 > gpurun_out/r4b1/c3_m1.json 2> gpurun_out/r4b1/c3_m1.err
RVT_HCX=0 $B > gpurun_out/r4b1/c3_hcw.json 2> gpurun_out/r4b1/c3_hcw.err
for f in c3 c3_m0 c3_m1 c3_hcw; do python - gpurun_out/r4b1/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['avg_launch_ms'], d['config'].get('genes_handed_back_per_step'), d.get('kernel_time_share',{}).get('device_ms_per_step'), d.get('parity',{}).get('p_max_abs_diff'))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
