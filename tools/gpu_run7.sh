#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "step_up or generation2" > gpurun_out/t_big.log 2>&1
echo "pytest exit $?" > gpurun_out/summary.log; tail -30 gpurun_out/t_big.log | cut -c1-300 >> gpurun_out/summary.log
cat gpurun_out/summary.log
