#!/bin/bash
# A/B of two library builds on a GPU box: tools/bin/libdsdgp_base.so (a copy of the previous build) against csrc/libdsdgp.so (working
# tree).  Parity subset first (working tree), then tools/ab_kernels.py per config, interleaved twice.  AB_CFGS="2 3", PYTEST_K="..."
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/ab; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "${PYTEST_K:-grad or parity or elbo or cfg or M100 or non_power or golden}" > $O/t_sel.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; grep -E "passed|failed|Error" $O/t_sel.log | tail -3 >> $O/summary.log
for rep in 1 2; do
  for lib in tools/bin/libdsdgp_base.so doubly-stochastic-dgp_amd/csrc/libdsdgp.so; do
    echo "== $lib" >> $O/summary.log
    DSDGP_LIB_PATH=$R/$lib timeout 400 python tools/ab_kernels.py ${AB_CFGS:-2 3} 2>&1 | grep "^{" >> $O/summary.log
  done
done
cat $O/summary.log
