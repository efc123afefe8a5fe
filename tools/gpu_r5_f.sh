#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r5f; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "full_size or cfg4 or cfg5 or large_M or more_than_1024 or mp_2048 or natgrad" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/summary.log; grep "passed\|failed" $O/pytest.log >> $O/summary.log
for i in 1 2; do timeout 300 python tools/ab_kernels.py 4 5 2>&1 | grep "^{" >> $O/summary.log; done
cat $O/summary.log
