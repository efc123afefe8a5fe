#!/bin/bash
# round 6, batch j: side stream priority (DSDGP_SIDE_PRIO = -1 high / 1 low / unset normal), cfg 2 3, interleaved twice
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6j; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
  for p in none -1 1; do
    echo "== side prio $p" >> $O/summary.log
    if [ $p = none ]; then unset DSDGP_SIDE_PRIO; else export DSDGP_SIDE_PRIO=$p; fi
    timeout 400 python tools/ab_kernels.py 2 3 2>&1 | grep "^{\|side prio" >> $O/summary.log
  done
done
cat $O/summary.log
