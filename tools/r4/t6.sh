#!/bin/bash
T="tests/test_gpu_hardcall.py tests/test_gpu_parity.py tests/test_gpu_stream.py"
for E in "RVT_HCX=0" "RVT_HCX_FUSED=0"; do
  echo "== $E"; env $E timeout 900 python -m pytest $T -q -x 2>&1 | grep -E "^E |Error|assert|FAILED|test_" | head -20
done
