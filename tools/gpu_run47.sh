#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t47.log 2>&1; grep -n "passed\|failed" gpurun_out/t47.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-130; done
timeout 600 python tools/bench_configs.py 1 3 4 5 2>&1 | tail -4 | cut -c1-150
