#!/bin/bash
mkdir -p gpurun_out/r4b19
run() { tag=$1; shift; env "$@" > gpurun_out/r4b19/$tag.json 2> gpurun_out/r4b19/$tag.err; python - gpurun_out/r4b19/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],3), d['roofline']['traffic'], round(d['kernel_time_share']['device_ms_per_step'],1), d.get('cpu_baseline',{}).get('value'))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
run qt python bench.py
run c3 python bench.py --trait binary --samples 200000 --no-cpu-baseline --no-from-host
run c1 python bench.py --no-cpu-baseline --no-from-host --samples 50000 --m-lo 30 --m-hi 30 --genes 1024 --tests 1
run dosage python bench.py --dosage --no-cpu-baseline --no-from-host
