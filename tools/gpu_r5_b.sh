#!/bin/bash
# round 5, batch B: Gram launch geometry A/B (+ its tests), Csave policy A/B at cfg 2 / 3
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r5b; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "gram" > $O/pytest.log 2>&1; echo "pytest gram rc=$?" > $O/summary.log; grep "passed\|failed" $O/pytest.log >> $O/summary.log
for v in 1 2 1 2; do echo "== gram v$v" >> $O/summary.log; DSDGP_GRAM_V=$v timeout 120 python tools/gram_time.py 2>&1 | grep "n=" >> $O/summary.log; done
echo "== csave cfg2" >> $O/summary.log
timeout 300 python tools/ab_force.py 2 "save_c=1" "save_c=2,cs_min_dout=1,cs_max_dout=1" "save_c=2,cs_min_dout=1" 2>&1 | grep "^{" >> $O/summary.log
echo "== csave cfg3" >> $O/summary.log
timeout 300 python tools/ab_force.py 3 "save_c=1" "save_c=2,cs_min_dout=1" "save_c=2,cs_min_dout=1,cs_max_dout=1" 2>&1 | grep "^{" >> $O/summary.log
cat $O/summary.log
