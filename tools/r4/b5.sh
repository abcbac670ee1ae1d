#!/bin/bash
B="python bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --missing-frac 0"
mkdir -p gpurun_out/r4b5
run() { tag=$1; shift; env "$@" $B > gpurun_out/r4b5/$tag.json 2> gpurun_out/r4b5/$tag.err; python - gpurun_out/r4b5/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d['roofline']['avg_launch_ms'], d.get('kernel_time_share',{}).get('device_ms_per_step'), d.get('parity',{}).get('p_max_abs_diff'))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run base64 RVT_PV_CUS=64
run w3_64 RVT_LIBRARY=rvtests_amd/csrc/librvtests_amd_pv3.so RVT_PV_CUS=64
run w3_32 RVT_LIBRARY=rvtests_amd/csrc/librvtests_amd_pv3.so RVT_PV_CUS=32
run w4_64 RVT_LIBRARY=rvtests_amd/csrc/librvtests_amd_pv4.so RVT_PV_CUS=64
run w4_32 RVT_LIBRARY=rvtests_amd/csrc/librvtests_amd_pv4.so RVT_PV_CUS=32
