#!/bin/bash
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d.get('kernel_time_share',{}).get('device_ms_per_step'))"; }
B="python bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host"
for w in 64 96 128 192; do echo "== config3 WPARTS=$w"; RVT_WPARTS=$w $B 2>&1 | line; done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host"
for w in 64 96 128; do echo "== default WPARTS=$w"; RVT_WPARTS=$w $B 2>&1 | line; done
