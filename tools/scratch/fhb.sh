#!/bin/bash
# fp64 pageable hand-off inside the bench process: runtime path vs staged ring (scratch)
for big in 0 1; do for thr in 3 6; do
RVT_STAGE_BIG=$big RVT_COPY_THREADS=$thr python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('big=$big thr=$thr', {k:round(v.get('gene_sets_per_s',0)) for k,v in d['from_host'].items() if k.startswith('fp64')})"
done; done
