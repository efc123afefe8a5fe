#!/bin/bash
# full GPU suite, then A/B of the grouped-GEMM fast path (libdsdgp_head.so = the same tree with HEAD's linalg.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2p; mkdir -p $O; rm -f $O/summary.log
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; tail -3 $O/t_all.log >> $O/summary.log
for lib in libdsdgp_head.so libdsdgp.so; do
  echo "== $lib" >> $O/summary.log
  DSDGP_LIB_PATH=$R/doubly-stochastic-dgp_amd/csrc/$lib timeout 400 python tools/ab_kernels.py 2 3 4 5 2>&1 | grep "^{" >> $O/summary.log
done
cat $O/summary.log
