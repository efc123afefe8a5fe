#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2l; mkdir -p $O; rm -f $O/summary.log
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -15 $O/pytest.log | cut -c1-300 >> $O/summary.log
timeout 300 python tools/bench_configs.py 1 3 4 5 2>&1 | grep config | cut -c1-200 >> $O/summary.log
timeout 300 python tools/ab_kernels.py 2 2>&1 | grep cfg >> $O/summary.log
cat $O/summary.log
