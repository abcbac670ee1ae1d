#!/bin/bash
T="tests/test_gpu_hardcall.py tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_lattice.py tests/test_gpu_bgen.py tests/test_gpu_edge.py"
for E in "RVT_HCX=0" "RVT_HCX_FUSED=0" "RVT_PV_CUS=0" "RVT_FDX=0" "RVT_PV_CUS=32 RVT_AS_THREADS=256"; do
  echo "== $E"; env $E timeout 900 python -m pytest $T -q -x 2>&1 | tail -2
done
echo "== RVT_FDX=0 floatdosage (expected: the 'take it' assertions fail, records still equal)"; RVT_FDX=0 timeout 300 python -m pytest tests/test_gpu_floatdosage.py -q 2>&1 | tail -3
