#!/bin/bash
# round 6, batch d: 4 s + g distances + k_asm_rows loads up front (parity, A/B), shard steps with the fused last layer on / off
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6d; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; tail -4 $O/t_all.log >> $O/summary.log
for rep in 1 2; do
  echo "== base (before the fused last layer)" >> $O/summary.log
  DSDGP_LIB_PATH=$R/tools/bin/libdsdgp_base.so timeout 400 python tools/ab_kernels.py 2 2>&1 | grep "^{" >> $O/summary.log
  echo "== tree" >> $O/summary.log
  timeout 400 python tools/ab_kernels.py 2 1 3 2>&1 | grep "^{" >> $O/summary.log
  echo "== tree asm_pre=0" >> $O/summary.log
  DSDGP_FORCE=asm_pre=0 timeout 400 python tools/ab_kernels.py 2 1 2>&1 | grep "^{" >> $O/summary.log
done
echo "== shards, tree" >> $O/summary.log
timeout 600 python tools/bench_shards.py 2>&1 | grep "^{" >> $O/summary.log
echo "== shards, last_fuse=0" >> $O/summary.log
DSDGP_FORCE=last_fuse=0 timeout 600 python tools/bench_shards.py 2>&1 | grep "^{" >> $O/summary.log
cat $O/summary.log
