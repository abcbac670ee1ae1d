#!/bin/bash
B="python bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host"
mkdir -p gpurun_out/r4b6
run() { tag=$1; shift; $B "$@" > gpurun_out/r4b6/$tag.json 2> gpurun_out/r4b6/$tag.err; python - gpurun_out/r4b6/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d['roofline']['avg_launch_ms'], d['config'].get('genes_handed_back_per_step'), d.get('kernel_time_share',{}).get('device_ms_per_step'))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
run m0 --missing-frac 0
run m05
run m1 --missing-frac 1.0
timeout 900 python -m pytest tests/test_gpu_hardcall.py tests/test_gpu_parity.py tests/test_gpu_stream.py -q 2>&1 | tail -5
