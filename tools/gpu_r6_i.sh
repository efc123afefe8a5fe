#!/bin/bash
# round 6, batch i: fork events on the chains' own dispatches (ext_ev = 2) and the flag join (flag_join = 1): parity subset, then A/B of the
# four combinations on one library, interleaved twice; shards at 125 rows
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6i; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
DSDGP_FORCE=ext_ev=2,flag_join=1 timeout 900 python -m pytest tests -m gpu -q -x -k "overlap or golden or full_size or adam or minibatch or train" > $O/t_sel.log 2>&1; echo "pytest(ext_ev=2,flag_join=1) rc=$?" >> $O/summary.log; tail -3 $O/t_sel.log >> $O/summary.log
for rep in 1 2; do
  for f in "ext_ev=1,flag_join=0" "ext_ev=2,flag_join=0" "ext_ev=1,flag_join=1" "ext_ev=2,flag_join=1"; do
    echo "== $f" >> $O/summary.log
    DSDGP_FORCE=$f timeout 400 python tools/ab_kernels.py 2 2>&1 | grep "^{" >> $O/summary.log
  done
done
for f in "ext_ev=1,flag_join=0" "ext_ev=2,flag_join=1"; do
  echo "== $f cfg 3 4 5" >> $O/summary.log
  DSDGP_FORCE=$f timeout 600 python tools/ab_kernels.py 3 4 5 2>&1 | grep "^{" >> $O/summary.log
done
cd /tmp
for f in "ext_ev=1,flag_join=0" "ext_ev=2,flag_join=1"; do
  rm -rf /tmp/tl
  DSDGP_FORCE=$f timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $R/tools/shard_timeline.py 1000 > $O/run_$f.log 2>&1
  DB=$(find /tmp/tl -name "*.db" | head -1)
  python $R/tools/timeline_dump.py $DB k_tail 3 > "$O/step_$f.txt"
done
cat $O/summary.log $O/step_*.txt
