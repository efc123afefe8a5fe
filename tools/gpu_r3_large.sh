#!/bin/bash
# rocprof kernel stats of the large-M config shapes (cfg 4: M = 512, cfg 5: M = 1024 + natgrad), serial schedule
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r3large; mkdir -p $O; export TMPDIR=/tmp
for c in ${CFGS:-4 5}; do
  rm -rf /tmp/prof$c
  (cd /tmp && DSDGP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$c -o p -- python $R/tools/ab_kernels.py $c > $O/run$c.log 2>&1)
  DB=$(find /tmp/prof$c -name "*results.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $O/stats$c.md "cfg $c shape, tools/ab_kernels.py under rocprofv3 --kernel-trace --stats, serial schedule" > /dev/null
done
ls $O
