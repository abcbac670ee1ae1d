#!/bin/bash
./tools/fdx_bench check | tail -12
for s in 1 2 3 5 9 10; do
  echo "== PYTHONHASHSEED=$s"; PYTHONHASHSEED=$s timeout 300 python -m pytest "tests/test_gpu_stream.py::test_long_stream_equals_block_submission[bgen]" -q -x 2>&1 | grep -E "AssertionError: |passed|failed" | head -3
done
timeout 600 python -m pytest tests/test_gpu_floatdosage.py tests/test_gpu_bgen.py -q -x 2>&1 | tail -3
