#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r3head; mkdir -p $O; export TMPDIR=/tmp
DSDGP_FORCE="head=1" timeout 300 python tools/potrf_timing.py 2>&1 | grep cycles > $O/summary.log
DSDGP_FORCE="head=0" timeout 300 python tools/potrf_timing.py 2>&1 | grep cycles >> $O/summary.log
timeout 600 python -m pytest tests -m gpu -q -x -k "potrf or conditional or elbo_value or gradients_three or white or ill or natgrad or ragged" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; tail -3 $O/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" >> $O/summary.log
timeout 300 python tools/ab_force.py 2 "head=0" "head=1" 2>/dev/null >> $O/summary.log
cat $O/summary.log
