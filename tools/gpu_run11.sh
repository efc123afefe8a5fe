#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "potrf or cfg5 or cfg4 or large_M or natgrad" 2>&1 | tail -15
timeout 600 python tools/bench_configs.py 4 5 2>&1 | tail -4
