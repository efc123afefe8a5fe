#!/bin/bash
for v in scalar d4; do
  if [ $v = scalar ]; then export DSDGP_LIB_PATH=$PWD/tools/bin/libdsdgp_scalar.so; else unset DSDGP_LIB_PATH; fi
  echo "== variant $v"
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gradient or white or cfg3 or full_size" 2>&1 | tail -2
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-130
  timeout 600 python tools/bench_configs.py 3 4 5 2>&1 | tail -3 | cut -c1-150
done
