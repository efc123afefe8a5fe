#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r3potrf; mkdir -p $O; export TMPDIR=/tmp
rm -rf /tmp/pp; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python $R/tools/potrf_prof.py 1024 > $O/run.log 2>&1)
DB=$(find /tmp/pp -name "*results.db" | head -1)
python $R/tools/rocprof_summary.py $DB $O/stats.md "dsdgp_potrf n = 1024, 6 calls" > /dev/null
grep "potrf n=\|relerr" $O/run.log; head -16 $O/stats.md | cut -c1-150
