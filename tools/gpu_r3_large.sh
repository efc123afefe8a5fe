#!/bin/bash
# per-kernel event times of the large-M config shapes (cfg 4: M = 512, cfg 5: M = 1024 + natgrad) + rocprof kernel stats of cfg 5
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3large; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/ab.log
for c in 4 5; do timeout 300 python tools/ab_kernels.py $c 2>&1 | grep "^{" >> $O/ab.log; done
for c in 4 5; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/../../$O/prof$c -o p -- python $GRAFT_REPO_ROOT/tools/ab_kernels.py $c > /dev/null 2>&1)
  f=$(find $O/prof$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -25 "$f" | cut -c1-200 > $O/stats$c.csv
  rm -rf $O/prof$c
done
cat $O/ab.log
