#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_floatdosage.py -q -x 2>&1 | tail -15
python bench.py --dosage --dosage-float --no-cpu-baseline --no-from-host > gpurun_out/bench_fdx.json 2> gpurun_out/bench_fdx.err; tail -c 2500 gpurun_out/bench_fdx.json | head -c 1500; tail -3 gpurun_out/bench_fdx.err
