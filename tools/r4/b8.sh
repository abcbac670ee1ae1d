#!/bin/bash
mkdir -p gpurun_out/r4b8
B="python bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host"
run() { tag=$1; shift; env "$@" $B > gpurun_out/r4b8/$tag.json 2> gpurun_out/r4b8/$tag.err; python - gpurun_out/r4b8/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],3), d['roofline']['launches'], d.get('parity',{}).get('p_max_abs_diff'))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
run unfused64 RVT_HCX_FUSED=0 RVT_PV_CUS=64
run fused64 RVT_PV_CUS=64
run fused48 RVT_PV_CUS=48
run fused40 RVT_PV_CUS=40
run fused32 RVT_PV_CUS=32
run fused56 RVT_PV_CUS=56
