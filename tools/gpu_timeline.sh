#!/bin/bash
# one training step's kernel timeline of config 2 in the production (overlapped) schedule: gpurun_out/timeline/{step,gaps}.txt
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/timeline; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tl -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/prod.json 2> $O/prod.err
DB=$(find /tmp/tl -name "*.db" | head -1)
python $R/tools/gap_analysis.py $DB k_tail > $O/gaps.txt
python $R/tools/timeline_dump.py $DB k_tail 3 > $O/step.txt
cat $O/step.txt
