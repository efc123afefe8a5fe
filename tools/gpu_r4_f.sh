#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; mkdir -p gpurun_out; O=$R/gpurun_out/r4f; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; grep -n "passed\|failed" $O/t_all.log; grep -n "^E " $O/t_all.log | head -8
timeout 600 python tools/bench_configs.py 4 5 2>&1 | grep "^{" | cut -c1-200
timeout 600 python tools/bench_configs.py 5 2>&1 | grep "^{" | cut -c1-200
