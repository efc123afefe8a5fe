#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r3head; mkdir -p $O; export TMPDIR=/tmp
DSDGP_FORCE="head=1" timeout 300 python tools/potrf_timing.py 2>&1 | grep cycles > $O/summary.log
timeout 300 python tools/ab_force.py 2 "head=0" "head=1" 2>/dev/null >> $O/summary.log
cat $O/summary.log
