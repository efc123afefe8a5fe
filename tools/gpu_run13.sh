#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 300 python bench.py 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py 2>&1 | tail -1 | cut -c1-200
timeout 600 python tools/bench_configs.py 1 3 4 5 2>&1 | tail -4
