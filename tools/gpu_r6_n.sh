#!/bin/bash
# round 6, batch n: the first layer's forward chain inside the head launch (k_head_fwd): new tests first (bounded), then the suite, A/B, timelines
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6n; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -k "head_launch" > $O/t_new.log 2>&1; echo "pytest(new) rc=$?" >> $O/summary.log; tail -15 $O/t_new.log >> $O/summary.log
if grep -q "rc=0" $O/summary.log; then
  timeout 1500 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "pytest(all) rc=$?" >> $O/summary.log; tail -4 $O/t_all.log >> $O/summary.log
fi
for rep in 1 2; do
  for f in "head_fwd=0" "head_fwd=1"; do
    echo "== $f" >> $O/summary.log
    DSDGP_FORCE=$f timeout 400 python tools/ab_kernels.py 2 2>&1 | grep "^{" >> $O/summary.log
  done
done
echo "== shards (tree)" >> $O/summary.log
timeout 600 python tools/bench_shards.py 2>&1 | grep "^{" | cut -c1-200 >> $O/summary.log
cd /tmp
for rows in 1000 125; do
  rm -rf /tmp/tl$rows
  timeout 300 rocprofv3 --kernel-trace -d /tmp/tl$rows -o t -- python $R/tools/shard_timeline.py $rows > $O/run$rows.log 2>&1
  DB=$(find /tmp/tl$rows -name "*.db" | head -1)
  python $R/tools/timeline_dump.py $DB k_tail 3 > $O/step_$rows.txt
done
cat $O/summary.log $O/step_1000.txt $O/step_125.txt
