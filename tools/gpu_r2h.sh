#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2h; mkdir -p $O; rm -f $O/summary.log
export TMPDIR=/tmp
timeout 300 python tools/gram_time.py >> $O/summary.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -k "gram" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -5 $O/pytest.log | cut -c1-300 >> $O/summary.log
cat $O/summary.log
