#!/bin/bash
B="python bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --missing-frac 0"
mkdir -p gpurun_out/r4b4
run() { tag=$1; shift; env "$@" $B > gpurun_out/r4b4/$tag.json 2> gpurun_out/r4b4/$tag.err; python - gpurun_out/r4b4/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d['roofline']['avg_launch_ms'], d.get('kernel_time_share',{}).get('device_ms_per_step'))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run pv64 RVT_PV_CUS=64
run pv64_t64 RVT_PV_CUS=64 RVT_TAIL_CUS=64
run pv64_t96 RVT_PV_CUS=64 RVT_TAIL_CUS=96
run pv64_t128 RVT_PV_CUS=64 RVT_TAIL_CUS=128
run pv32_t64 RVT_PV_CUS=32 RVT_TAIL_CUS=64
run pv32_t96 RVT_PV_CUS=32 RVT_TAIL_CUS=96
run pv96_t96 RVT_PV_CUS=96 RVT_TAIL_CUS=96
