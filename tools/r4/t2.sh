#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_hardcall.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_edge.py tests/test_gpu_stream.py tests/test_gpu_null.py tests/test_gpu_group.py tests/test_gpu_metascore.py -q 2>&1 | tail -40
