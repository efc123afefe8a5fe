#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for t in 256 384 512 768 1024; do
  echo "target $t"; DSDGP_WGRAD_TARGET=$t timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-130
done
for t in 512 1024; do
  echo "cfg3 target $t"; DSDGP_WGRAD_TARGET=$t timeout 300 python tools/bench_configs.py 3 2>&1 | tail -1
done
