#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r3c6; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -4 $O/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" >> $O/summary.log; grep -E "^E |Error|FAILED" $O/pytest.log | head -10 >> $O/summary.log
timeout 600 python tools/ab_force.py 2 ${AB:-"adj_fuse=0" "adj_fuse=1"} > $O/ab_force.log 2> $O/ab_force.err; cat $O/ab_force.log >> $O/summary.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/prof -o t -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/prof.json 2> $O/prof.err
db=$(find $O/prof -name "*.db" | head -1)
python $R/tools/timeline_dump.py $db k_tail 3 > $O/timeline.txt 2>&1
python $R/tools/gap_analysis.py $db k_tail > $O/gap.txt 2>&1
rm -rf $O/prof
cd $R
cat $O/summary.log
