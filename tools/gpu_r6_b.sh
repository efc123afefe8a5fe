#!/bin/bash
# round 6, batch b: fused last-layer launch — parity (round-6 tests, full-size fixtures, golden), A/B against the base library and with
# the launch switched off, per-kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6b; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_full_size.py tests/test_golden.py tests/test_gpu_parity.py -m gpu -q -x > $O/t_sel.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; tail -8 $O/t_sel.log >> $O/summary.log
for rep in 1 2; do
  echo "== base" >> $O/summary.log
  DSDGP_LIB_PATH=$R/tools/bin/libdsdgp_base.so timeout 400 python tools/ab_kernels.py ${AB_CFGS:-2 3} 2>&1 | grep "^{" >> $O/summary.log
  echo "== tree" >> $O/summary.log
  timeout 400 python tools/ab_kernels.py ${AB_CFGS:-2 3} 2>&1 | grep "^{" >> $O/summary.log
  echo "== tree last_fuse=0" >> $O/summary.log
  DSDGP_FORCE=last_fuse=0 timeout 400 python tools/ab_kernels.py ${AB_CFGS:-2 3} 2>&1 | grep "^{" >> $O/summary.log
done
cat $O/summary.log
