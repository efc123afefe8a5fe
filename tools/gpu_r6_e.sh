#!/bin/bash
# round 6, batch e: one step's kernel timeline at 1000 rows and at the 125-row shard (production schedule)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6e; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for rows in 1000 125; do
  rm -rf /tmp/tl$rows
  timeout 300 rocprofv3 --kernel-trace -d /tmp/tl$rows -o t -- python $R/tools/shard_timeline.py $rows > $O/run$rows.log 2>&1
  DB=$(find /tmp/tl$rows -name "*.db" | head -1)
  python $R/tools/timeline_dump.py $DB k_tail 3 > $O/step_$rows.txt
  python $R/tools/gap_analysis.py $DB k_tail > $O/gaps_$rows.txt
done
cat $O/step_1000.txt $O/step_125.txt
