#!/bin/bash
# round 6, batch f: Ku^-1-only event in front of the fused last layer; overlap threshold at the shard sizes
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6f; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -x > $O/t_sel.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; tail -4 $O/t_sel.log >> $O/summary.log
for rep in 1 2; do
  echo "== base (round 5)" >> $O/summary.log
  DSDGP_LIB_PATH=$R/tools/bin/libdsdgp_base.so timeout 400 python tools/ab_kernels.py 2 2>&1 | grep "^{" >> $O/summary.log
  echo "== tree" >> $O/summary.log
  timeout 400 python tools/ab_kernels.py 2 3 2>&1 | grep "^{" >> $O/summary.log
done
for om in 1048576 262144 65536; do
  echo "== shards, overlap_min=$om" >> $O/summary.log
  DSDGP_FORCE=overlap_min=$om timeout 600 python tools/bench_shards.py 2>&1 | grep "^{" | cut -c1-200 >> $O/summary.log
done
cd /tmp
for rows in 1000 125; do
  rm -rf /tmp/tl$rows
  DSDGP_FORCE=overlap_min=262144 timeout 300 rocprofv3 --kernel-trace -d /tmp/tl$rows -o t -- python $R/tools/shard_timeline.py $rows > $O/run$rows.log 2>&1
  DB=$(find /tmp/tl$rows -name "*.db" | head -1)
  python $R/tools/timeline_dump.py $DB k_tail 3 > $O/step_$rows.txt
done
cat $O/summary.log $O/step_1000.txt $O/step_125.txt
