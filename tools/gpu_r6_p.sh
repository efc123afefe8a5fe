#!/bin/bash
# round 6, batch p: head launch — the mirror blocks of the Z distances stored along rows (swapped-operand MFMA): parity subset, phase clocks, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6p; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "golden or full_size or parity or round6 or jitter or potrf" > $O/t_sel.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; tail -3 $O/t_sel.log >> $O/summary.log
echo "== phase clocks prev / tree" >> $O/summary.log
DSDGP_LIB_PATH=$R/tools/bin/libdsdgp_prev.so timeout 200 python tools/potrf_timing.py 2>&1 | grep "head cycles" | tail -2 >> $O/summary.log
timeout 200 python tools/potrf_timing.py 2>&1 | grep "head cycles" | tail -2 >> $O/summary.log
for rep in 1 2; do
  echo "== prev" >> $O/summary.log
  DSDGP_LIB_PATH=$R/tools/bin/libdsdgp_prev.so timeout 400 python tools/ab_kernels.py 2 2>&1 | grep "^{" >> $O/summary.log
  echo "== tree" >> $O/summary.log
  timeout 400 python tools/ab_kernels.py 2 2>&1 | grep "^{" >> $O/summary.log
done
cat $O/summary.log
