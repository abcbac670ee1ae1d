#!/bin/bash
mkdir -p gpurun_out/r4b12
run() { tag=$1; shift; env "$@" > gpurun_out/r4b12/$tag.json 2> gpurun_out/r4b12/$tag.err; python - gpurun_out/r4b12/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],3), round(d['kernel_time_share']['device_ms_per_step'],1), d.get('parity',{}).get('p_max_abs_diff'))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
C3="python bench.py --trait binary --samples 200000 --no-cpu-baseline --no-from-host"
run c3_pv64 $C3
run c3_pv56 RVT_PV_CUS=56 $C3
run c3_pv48 RVT_PV_CUS=48 $C3
run qt python bench.py --no-cpu-baseline --no-from-host
python3 tools/pv_prof.py --samples 200000 --genes 512 2>&1 | grep "pvalue kernel"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_edge.py tests/test_gpu_hardcall.py -q -x 2>&1 | tail -3
