#!/bin/bash
# one training step's kernel timeline in the production (overlapped) schedule: gpurun_out/timeline/{step,gaps}_cfg$C.txt
# usage: bash tools/gpu_timeline.sh [config id, default 2]
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
C=${1:-2}
O=$R/gpurun_out/timeline; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/tl
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tl -o t -- python $R/tools/ab_kernels.py $C > $O/prod_cfg$C.log 2>&1
DB=$(find /tmp/tl -name "*.db" | head -1)
python $R/tools/gap_analysis.py $DB k_tail > $O/gaps_cfg$C.txt
python $R/tools/timeline_dump.py $DB k_tail 6 > $O/step_cfg$C.txt
cat $O/step_cfg$C.txt
