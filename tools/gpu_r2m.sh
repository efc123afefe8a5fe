#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2m; mkdir -p $O; rm -f $O/summary.log
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -6 $O/pytest.log | cut -c1-300 >> $O/summary.log
timeout 600 python tools/bench_padding.py >> $O/summary.log 2>&1
cat $O/summary.log
