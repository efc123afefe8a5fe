#!/bin/bash
# round-2 GPU batch B: parity with block-major Csave; wgrad tile / split sweep
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
run() { echo "== $*" >> $O/ab.log; env "$@" timeout 300 python tools/ab_kernels.py 2 3 >> $O/ab.log 2>&1; }
run DSDGP_SAVE_C=1
run DSDGP_SAVE_C=0
run DSDGP_ALG_G=0
run DSDGP_WGRAD_NI=2 DSDGP_WGRAD_TARGET=2048
run DSDGP_WGRAD_NI=2 DSDGP_WGRAD_TARGET=4096
run DSDGP_WGRAD_NI=4 DSDGP_WGRAD_TARGET=2048
run DSDGP_WGRAD_NI=4 DSDGP_WGRAD_TARGET=4096
for v in "DSDGP_SAVE_C=1" "DSDGP_WGRAD_NI=2 DSDGP_WGRAD_TARGET=4096"; do
  echo "== $v" >> $O/ab.log
  env $v timeout 300 python tools/bench_configs.py 4 5 >> $O/ab.log 2>&1
done
tail -5 $O/pytest.log | tee -a $O/summary.log
grep -E "^FAILED|^ERROR" $O/pytest.log | head -20 | tee -a $O/summary.log
grep -E "==|cfg|config" $O/ab.log | cut -c1-260 | tee -a $O/summary.log
