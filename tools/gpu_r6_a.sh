#!/bin/bash
# round 6, batch a: the round-6 parity tests, the whole GPU suite, and the base library's per-kernel times at configs 2 / 3
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6a; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round6.py -m gpu -q -x > $O/t_r6.log 2>&1; echo "r6 pytest rc=$?" >> $O/summary.log; tail -15 $O/t_r6.log >> $O/summary.log
timeout 1200 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "all pytest rc=$?" >> $O/summary.log; tail -5 $O/t_all.log >> $O/summary.log
for rep in 1 2; do
  DSDGP_LIB_PATH=$R/tools/bin/libdsdgp_base.so timeout 400 python tools/ab_kernels.py 2 3 2>&1 | grep "^{" >> $O/summary.log
done
cat $O/summary.log
