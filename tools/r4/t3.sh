#!/bin/bash
timeout 1800 python -m pytest tests/test_gpu_stream.py tests/test_gpu_group.py tests/test_gpu_packed.py tests/test_gpu_perm.py tests/test_gpu_lattice.py tests/test_host_driver.py -q -x 2>&1 | tail -8
