#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t41.log 2>&1; grep -n "passed\|failed" gpurun_out/t41.log; grep -n "Error\|assert" gpurun_out/t41.log | head -5
timeout 300 python tools/ab_kernels.py 2 2>&1 | grep "^{"
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-130; done
bash tools/gpu_potrf_timing.sh | tail -2
