#!/bin/bash
./tools/host_feed_bench --registered --modes bed,int8 2>&1 | cut -c1-170
./tools/host_feed_bench --modes bed 2>&1 | cut -c1-170
timeout 600 python -m pytest tests/test_gpu_packed.py tests/test_gpu_stream.py -q -x 2>&1 | tail -3
