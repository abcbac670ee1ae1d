#!/bin/bash
mkdir -p gpurun_out/r4b7
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host > gpurun_out/r4b7/$tag.json 2> gpurun_out/r4b7/$tag.err; python - gpurun_out/r4b7/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d['roofline']['avg_launch_ms'], d.get('kernel_time_share',{}).get('device_ms_per_step'), d.get('parity',{}).get('p_max_abs_diff'))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
run qt_pv0 RVT_PV_CUS=0
run qt_pv64 RVT_PV_CUS=64
run qt_pv32 RVT_PV_CUS=32
