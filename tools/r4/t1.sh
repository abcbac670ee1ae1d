#!/bin/bash
timeout 300 ./tools/hcx_bench check
timeout 900 python -m pytest tests/test_gpu_hardcall.py -x -q 2>&1 | tail -15
