#!/bin/bash
# configs[3] shape: what the hand-back and the per-gene stages cost
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d.get('kernel_time_share',{}).get('device_ms_per_step'))"; }
B="python bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host"
echo "== all tests, 5% imputed"; $B 2>&1 | line
echo "== all tests, no imputed"; $B --missing-frac 0 2>&1 | line
echo "== skat only, no imputed"; $B --missing-frac 0 --tests 1 2>&1 | line
echo "== skato only, no imputed"; $B --missing-frac 0 --tests 2 2>&1 | line
echo "== burden only, no imputed"; $B --missing-frac 0 --tests 12 2>&1 | line
echo "== all, no imputed, WPARTS 32"; RVT_WPARTS=32 $B --missing-frac 0 2>&1 | line
echo "== all, no imputed, 1024 genes"; $B --missing-frac 0 --genes 1024 2>&1 | line
