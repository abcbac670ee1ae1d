#!/bin/bash
# round 4 probe: p-value kernel phase breakdown (profiling build) + kernel timeline of configs[3]
export TMPDIR=/tmp
mkdir -p gpurun_out/r4p1
RVT_LIBRARY=rvtests_amd/csrc/librvtests_amd_prof.so python3 tools/pv_prof.py --samples 50000 --genes 512 > gpurun_out/r4p1/pv_prof.txt 2>&1
bash tools/prof_trace.sh r4c3 --trait binary --samples 200000 --no-from-host > gpurun_out/r4p1/trace_c3.log 2>&1
python3 bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --missing-frac 0 --tests 12 > gpurun_out/r4p1/c3_burden_only.json 2> gpurun_out/r4p1/c3_burden_only.err
python3 bench.py --trait binary --samples 200000 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --missing-frac 0 > gpurun_out/r4p1/c3_nomiss.json 2> gpurun_out/r4p1/c3_nomiss.err
cat gpurun_out/r4p1/pv_prof.txt
