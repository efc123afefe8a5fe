#!/bin/bash
for v in old scalar d4; do
  if [ $v = d4 ]; then unset DSDGP_LIB_PATH; else export DSDGP_LIB_PATH=$PWD/tools/bin/libdsdgp_$v.so; fi
  echo "== variant $v"
  timeout 600 python tools/ab_kernels.py 2 3 4 5 2>&1 | grep "^{"
done
