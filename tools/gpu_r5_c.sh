#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r5c; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for v in 3; do DSDGP_GRAM_V=$v timeout 600 python -m pytest tests -m gpu -q -x -k "gram" > $O/pytest$v.log 2>&1; echo "pytest gram v$v rc=$?" >> $O/summary.log; grep "passed\|failed" $O/pytest$v.log >> $O/summary.log; done
for v in 2 3 2 3; do echo "== gram v$v" >> $O/summary.log; DSDGP_GRAM_V=$v timeout 120 python tools/gram_time.py 2>&1 | grep "n=" >> $O/summary.log; done
for occ in 4 5 8; do echo "== gram v3 occ $occ" >> $O/summary.log; DSDGP_GRAM_OCC=$occ DSDGP_GRAM_V=3 timeout 120 python tools/gram_time.py 2>&1 | grep "n=" >> $O/summary.log; done
cat $O/summary.log
