#!/bin/bash
for v in prev u8 wg2; do
  export DSDGP_LIB_PATH=$PWD/tools/bin/libdsdgp_$v.so
  echo "== variant $v"
  timeout 600 python tools/ab_kernels.py 2 2>&1 | grep "^{"
  timeout 600 python tools/ab_kernels.py 3 2>&1 | grep "^{"
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-130
done
