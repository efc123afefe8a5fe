#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r5g; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "gemm or full_size or cfg4 or cfg5 or large_M or more_than_1024 or mp_2048 or natgrad or potrf or trsm or golden" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/summary.log; grep "passed\|failed" $O/pytest.log >> $O/summary.log
for rep in 1 2; do
  for lib in tools/bin/libdsdgp_base.so doubly-stochastic-dgp_amd/csrc/libdsdgp.so; do
    echo "== $lib" >> $O/summary.log
    DSDGP_LIB_PATH=$R/$lib timeout 400 python tools/bench_configs.py 3 4 5 2>&1 | grep "^{" | cut -c1-120 >> $O/summary.log
  done
done
cat $O/summary.log
