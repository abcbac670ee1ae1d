#!/bin/bash
mkdir -p gpurun_out/r4b16
run() { tag=$1; shift; env "$@" > gpurun_out/r4b16/$tag.json 2> gpurun_out/r4b16/$tag.err; python - gpurun_out/r4b16/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],3), round(d['kernel_time_share']['device_ms_per_step'],1))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
C3="python bench.py --trait binary --samples 200000 --no-cpu-baseline --no-from-host"
QT="python bench.py --no-cpu-baseline --no-from-host"
run c3_default $C3
run c3_n500k python bench.py --trait binary --no-cpu-baseline --no-from-host
run qt_w64 $QT
run qt_w32 RVT_WPARTS=32 $QT
run qt_w48 RVT_WPARTS=48 $QT
timeout 900 python -m pytest tests/test_gpu_hardcall.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
