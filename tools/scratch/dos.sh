#!/bin/bash
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],2), r['kernel'], round(r['frac'],3), round(r['avg_launch_ms'],3), d['config'].get('hard_call_genes_per_step'), d['config'].get('genes_handed_back_per_step'), d.get('kernel_time_share',{}).get('device_ms_per_step'))"; }
B="python bench.py --dosage --steps 20 --warmup 5 --no-cpu-baseline --no-from-host"
echo "== lattice 1000"; $B 2>&1 | line
echo "== lattice not stated"; $B --dosage-lattice 0 2>&1 | line
echo "== default (hard calls)"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-host 2>&1 | line
