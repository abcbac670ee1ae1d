#!/bin/bash
# sweep of the CU partition (RVT_STAGE2_CUS) on the configs[3] shape and the default line
for cus in 0 32 64 96; do
  echo "== config3 RVT_STAGE2_CUS=$cus"
  RVT_STAGE2_CUS=$cus timeout 600 python bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d.get('kernel_time_share',{}).get('device_ms_per_step'))"
done
for cus in 0 32 64; do
  echo "== default RVT_STAGE2_CUS=$cus"
  RVT_STAGE2_CUS=$cus timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d.get('kernel_time_share',{}).get('device_ms_per_step'))"
done
