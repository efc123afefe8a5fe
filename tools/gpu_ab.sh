#!/bin/bash
# parity of the gradients, then A/B: libdsdgp_head.so (committed kernels) vs libdsdgp.so (working tree)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2q; mkdir -p $O; rm -f $O/summary.log
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "${PYTEST_K:-grad or parity or elbo or cfg or M100 or non_power}" > $O/t_sel.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; grep -E "passed|failed|Error" $O/t_sel.log | tail -3 >> $O/summary.log
for lib in libdsdgp_head.so libdsdgp.so; do
  echo "== $lib" >> $O/summary.log
  DSDGP_LIB_PATH=$R/doubly-stochastic-dgp_amd/csrc/$lib timeout 400 python tools/ab_kernels.py ${AB_CFGS:-2 3} 2>&1 | grep "^{" >> $O/summary.log
done
cat $O/summary.log
