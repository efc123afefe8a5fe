#!/bin/bash
# Platform probes of round 6 (stand-alone HIP programs; binaries are built here into tools/bin/, which is git-ignored):
#   xstream_probe : what a cross-stream dependency costs — event record / wait, an event on a kernel's completion signal, hipStreamWaitValue64
#                   on a flag the producer kernel writes (profiles/r06_xstream_probe.txt)
#   write_calib   : WRITE_SIZE / FETCH_SIZE on known byte counts in the chain kernels' store pattern (profiles/r06_write_size_calibration.txt)
#   xcd_probe     : which XCD a workgroup of a launch runs on
# Build first (no GPU needed):  for p in xstream_probe write_calib xcd_probe; do hipcc --offload-arch=gfx950 -O2 $([ $p = xstream_probe ] && echo -DPROBE_MAIN) tools/$p.hip -o tools/bin/$p; done
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/probes; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 120 $R/tools/bin/xstream_probe > $O/xstream_probe.txt 2>&1
timeout 60 $R/tools/bin/xcd_probe > $O/xcd_probe.txt 2>&1
cd /tmp
for ctr in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/wc_$ctr
  timeout 120 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/wc_$ctr -o p -- $R/tools/bin/write_calib > $O/run_$ctr.log 2>&1
  DB=$(find /tmp/wc_$ctr -name "*.db" | head -1)
  python - "$DB" $ctr >> $O/write_calib.txt <<'PY'
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
head -40 $O/xstream_probe.txt; cat $O/write_calib.txt
