#!/bin/bash
# round 6, batch k: WRITE_SIZE / FETCH_SIZE calibration on known byte counts (tools/write_calib.hip), per dispatch in launch order
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6k; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for ctr in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/wc_$ctr
  timeout 120 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/wc_$ctr -o p -- $R/tools/bin/write_calib > $O/run_$ctr.log 2>&1
  DB=$(find /tmp/wc_$ctr -name "*.db" | head -1)
  python - "$DB" $ctr >> $O/calib.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
key = "dispatch_id" if "dispatch_id" in cols else cols[0]
rows = cur.execute(f"select {key}, kernel_name, value from counters_collection where counter_name = ? order by {key}", (sys.argv[2],)).fetchall()
print(f"== {sys.argv[2]} per dispatch (KB -> MB), launch order; every kernel moves 20.48 MB")
for d, k, v in rows:
    print(f"  {d:5d}  {k.split('(')[0]:12s}  {v * 1024 / 1e6:8.2f} MB")
PY
done
cat $O/calib.txt
