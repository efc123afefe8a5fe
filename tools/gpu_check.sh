#!/bin/bash
# quick regression check on a GPU box: parity suite, per-kernel times, headline bench twice, the other config shapes
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t_check.log 2>&1; grep -n "passed\|failed" gpurun_out/t_check.log; grep -n "^E " gpurun_out/t_check.log | head -5
timeout 300 python tools/ab_kernels.py 2 2>&1 | grep "^{"
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['elbo_evals_per_s'], d['predict_f_rows_per_s'])"; done
timeout 600 python tools/bench_configs.py 1 3 4 5 2>&1 | tail -4 | cut -c1-150
