#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cfg3 or gradient or natgrad" 2>&1 | tail -2
timeout 600 python tools/ab_kernels.py 2 3 2>&1 | grep "^{"
