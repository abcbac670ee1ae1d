#!/bin/bash
# SQ counters of gene_pvalue_kernel on one isolated batch (tools/pv_prof.py, regular library)
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/pvpmc
mkdir -p $OUT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 tools/pv_prof.py --samples 200000 --genes 512 > $OUT/p$i.log 2>&1
  F=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  python3 - "$F" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'pvalue' not in k and 'spectrum' not in k and 'tridiag' not in k and 'assemble' not in k: continue
    k=k.split('(')[0][-40:]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
for k in acc:
    print(k, {c: v for c,v in acc[k].items()})
PY
  rm -rf $OUT/p$i
done
