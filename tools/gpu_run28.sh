#!/bin/bash
mkdir -p gpurun_out
for v in 0 1; do
  echo "== DSDGP_SM_SMALL=$v"
  DSDGP_SM_SMALL=$v timeout 600 python tools/ab_kernels.py 1 2>&1 | grep "^{"
  DSDGP_SM_SMALL=$v timeout 600 python tools/ab_kernels.py 2 2>&1 | grep "^{"
  DSDGP_SM_SMALL=$v timeout 600 python tools/ab_kernels.py 3 2>&1 | grep "^{"
done
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t28.log 2>&1; grep -n "passed\|failed" gpurun_out/t28.log
