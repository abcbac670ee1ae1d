#!/bin/bash
mkdir -p gpurun_out/r4b13
run() { tag=$1; shift; env "$@" > gpurun_out/r4b13/$tag.json 2> gpurun_out/r4b13/$tag.err; python - gpurun_out/r4b13/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],3), round(d['kernel_time_share']['device_ms_per_step'],1))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
QT="python bench.py --no-cpu-baseline --no-from-host"
run qt_base $QT
run qt_tail64 RVT_TAIL_CUS=64 $QT
run qt_tail96_pv64 RVT_TAIL_CUS=96 $QT
run qt_tail64_pv32 RVT_TAIL_CUS=64 RVT_PV_CUS=32 $QT
