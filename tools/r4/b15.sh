#!/bin/bash
mkdir -p gpurun_out/r4b15
run() { tag=$1; shift; env "$@" > gpurun_out/r4b15/$tag.json 2> gpurun_out/r4b15/$tag.err; python - gpurun_out/r4b15/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],3), round(d['kernel_time_share']['device_ms_per_step'],1))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
C3="python bench.py --trait binary --samples 200000 --no-cpu-baseline --no-from-host"
for w in 4 6 8 10 8 6; do run c3_w${w}_$RANDOM RVT_WPARTS=$w $C3; done
run c3_m1_w8 RVT_WPARTS=8 $C3 --missing-frac 1.0
run c3_m1_w16 $C3 --missing-frac 1.0
