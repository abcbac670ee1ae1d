#!/bin/bash
mkdir -p gpurun_out/r4b10
run() { tag=$1; shift; env "$@" > gpurun_out/r4b10/$tag.json 2> gpurun_out/r4b10/$tag.err; python - gpurun_out/r4b10/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],3), d['roofline']['launches'], round(d['kernel_time_share']['device_ms_per_step'],1))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
B="python bench.py --trait binary --samples 200000 --no-cpu-baseline --no-from-host"
run s20 $B --steps 20 --warmup 5
run s60 $B --steps 60 --warmup 8
run s60_unfused RVT_HCX_FUSED=0 $B --steps 60 --warmup 8
run s60_pv56 RVT_PV_CUS=56 $B --steps 60 --warmup 8
run s60_pv72 RVT_PV_CUS=72 $B --steps 60 --warmup 8
run qt40 python bench.py --no-cpu-baseline --no-from-host
run qt40_unfused RVT_HCX_FUSED=0 python bench.py --no-cpu-baseline --no-from-host
