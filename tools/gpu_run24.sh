#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t24.log 2>&1; tail -3 gpurun_out/t24.log
timeout 600 python tools/ab_kernels.py 2 3 4 5 2>&1 | grep "^{"
