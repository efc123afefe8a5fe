#!/bin/bash
# round 6, batch m: one-split plans write their tiles from registers (wg_red = 1 default) — full suite, then A/B against the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6m; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; tail -4 $O/t_all.log >> $O/summary.log
for rep in 1 2; do
  echo "== prev" >> $O/summary.log
  DSDGP_LIB_PATH=$R/tools/bin/libdsdgp_prev.so timeout 600 python tools/ab_kernels.py 2 4 5 2>&1 | grep "^{" >> $O/summary.log
  echo "== tree" >> $O/summary.log
  timeout 600 python tools/ab_kernels.py 2 4 5 2>&1 | grep "^{" >> $O/summary.log
done
timeout 600 python tools/bench_configs.py 3 4 5 2>/dev/null | grep "^{" | cut -c1-200 >> $O/summary.log
cat $O/summary.log
