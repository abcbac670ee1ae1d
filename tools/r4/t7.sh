#!/bin/bash
for s in 1 2 3 4 5 6 7 8; do
  echo "== PYTHONHASHSEED=$s"; PYTHONHASHSEED=$s timeout 300 python -m pytest "tests/test_gpu_stream.py::test_long_stream_equals_block_submission[bgen]" -q -x 2>&1 | grep -E "AssertionError: |passed|failed" | head -3
done
