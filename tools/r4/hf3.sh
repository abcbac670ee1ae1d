#!/bin/bash
./tools/host_feed_bench --registered --batch 8 --modes bed,int8 2>&1 | cut -c1-200
./tools/host_feed_bench --batch 8 --modes bed 2>&1 | cut -c1-200
./tools/host_feed_bench --registered --batch 16 --modes bed 2>&1 | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_packed.py tests/test_gpu_stream.py -q -x 2>&1 | tail -3
