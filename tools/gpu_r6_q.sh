#!/bin/bash
# round 6, batch q: the memory-bound passes of large models (parameter transforms without the constant zero tiles, Adam without the loads of
# masked pairs, four entries in flight in the q_sqrt gradient rows): full suite, A/B against the previous build, kernel stats of configs 4 / 5
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6q; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; tail -3 $O/t_all.log >> $O/summary.log
for rep in 1 2; do
  echo "== prev" >> $O/summary.log
  DSDGP_LIB_PATH=$R/tools/bin/libdsdgp_prev.so timeout 600 python tools/ab_kernels.py 2 4 5 2>&1 | grep "^{" | cut -c1-70 >> $O/summary.log
  echo "== tree" >> $O/summary.log
  timeout 600 python tools/ab_kernels.py 2 4 5 2>&1 | grep "^{" | cut -c1-70 >> $O/summary.log
done
for c in 4 5; do
  rm -rf /tmp/prof$c
  (cd /tmp && DSDGP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$c -o p -- python $R/tools/ab_kernels.py $c > $O/run$c.log 2>&1)
  DB=$(find /tmp/prof$c -name "*results.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $O/kernel_stats_cfg$c.md "config-$c" > /dev/null
  grep "k_adam\|k_prep_kuu\|k_asm\|k_tail\|k_reduce" $O/kernel_stats_cfg$c.md | cut -c1-120 >> $O/summary.log
done
cat $O/summary.log
