#!/bin/bash
# round 6, batch l: in-launch split-K reduction of the weight-gradient products (wg_red = 1) against the reduction launch (wg_red = 0)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6l; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; tail -4 $O/t_all.log >> $O/summary.log
for rep in 1 2; do
  for f in "wg_red=0" "wg_red=1"; do
    echo "== $f" >> $O/summary.log
    DSDGP_FORCE=$f timeout 400 python tools/ab_kernels.py 2 1 2>&1 | grep "^{" >> $O/summary.log
  done
done
for f in "wg_red=0" "wg_red=1"; do
  echo "== $f cfg 3 4 5" >> $O/summary.log
  DSDGP_FORCE=$f timeout 600 python tools/ab_kernels.py 3 4 5 2>&1 | grep "^{" >> $O/summary.log
done
cd /tmp
for f in "wg_red=1"; do
  rm -rf /tmp/tl
  DSDGP_FORCE=$f timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $R/tools/shard_timeline.py 1000 > $O/run_$f.log 2>&1
  DB=$(find /tmp/tl -name "*.db" | head -1)
  python $R/tools/timeline_dump.py $DB k_tail 3 > "$O/step_$f.txt"
done
cat $O/summary.log $O/step_*.txt
