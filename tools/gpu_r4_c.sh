#!/bin/bash
# per-launch table of the grouped M x M GEMM launches at config 5, 64 x 64 tiles (default) against 128 x 128 tiles from 1024
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; mkdir -p gpurun_out; O=$R/gpurun_out/r4c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for v in default big; do
  rm -rf /tmp/prof_$v
  if [ $v = big ]; then export DSDGP_GEMM_BIG_MIN=1024; fi
  (cd /tmp && DSDGP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o p -- python $R/tools/ab_kernels.py 5 > $O/run_$v.log 2>&1)
  DB=$(find /tmp/prof_$v -name "*results.db" | head -1)
  python $R/tools/launch_table.py $DB gemm_grouped gemm_big gemm_small chol trtri > $O/launches_$v.md
  echo "== $v"; grep "^{" $O/run_$v.log; cat $O/launches_$v.md | cut -c1-160
done
