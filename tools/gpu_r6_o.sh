#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6o; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for naps in 1 4 16; do
  echo "== naps $naps" >> $O/summary.log
  DSDGP_HEAD_FWD_NAPS=$naps DSDGP_HEAD_FWD_TIMING=1 timeout 120 python tools/shard_timeline.py 1000 2>&1 | grep "head+fwd" | tail -2 >> $O/summary.log
  DSDGP_HEAD_FWD_NAPS=$naps timeout 400 python tools/ab_kernels.py 2 2>&1 | grep "^{" | cut -c1-90 >> $O/summary.log
done
cat $O/summary.log
