#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r5d; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2 3; do for occ in 4 6 8 10 12 16 24; do echo "== v3 occ $occ" >> $O/summary.log; DSDGP_GRAM_OCC=$occ DSDGP_GRAM_V=3 timeout 120 python tools/gram_time.py 2>&1 | grep "n=1024\|n=512" >> $O/summary.log; done; done
cat $O/summary.log
