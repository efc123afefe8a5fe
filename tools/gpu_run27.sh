#!/bin/bash
mkdir -p gpurun_out
for v in old new; do
  if [ $v = new ]; then unset DSDGP_LIB_PATH; else export DSDGP_LIB_PATH=$PWD/tools/bin/libdsdgp_$v.so; fi
  echo "== variant $v"
  timeout 600 python tools/ab_kernels.py 1 2 3 4 5 2>&1 | grep "^{"
done
unset DSDGP_LIB_PATH
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t27.log 2>&1; grep -n "passed\|failed" gpurun_out/t27.log
