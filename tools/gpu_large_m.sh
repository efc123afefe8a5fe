#!/bin/bash
# M > 1024 shapes (tools/bench_configs.py x6 / x7): step rate + kernel stats and per-launch table of x6, dsdgp_potrf at n = 2048
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; P=$R/gpurun_out/large_m; rm -rf $P; mkdir -p $P
export TMPDIR=/tmp
timeout 600 python tools/bench_configs.py 6 7 > $P/r04_large_m_shapes.jsonl 2> $P/err.log
rm -rf /tmp/prof6
(cd /tmp && DSDGP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof6 -o p -- python $R/tools/ab_kernels.py 6 > $P/run6.log 2>&1)
DB=$(find /tmp/prof6 -name "*results.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $P/r04_kernel_stats_m2048.md "round 4: 3-layer M = 2048 shape (tools/ab_kernels.py 6) under rocprofv3 --kernel-trace --stats, serial schedule" > /dev/null
[ -n "$DB" ] && python $R/tools/launch_table.py $DB pgemm kuf thin gl_ wgrad gemm_grouped chol > $P/r04_launch_shapes_m2048.md
timeout 200 python tools/potrf_prof.py 2048 > $P/potrf2048.log 2>&1
cat $P/r04_large_m_shapes.jsonl; head -30 $P/r04_kernel_stats_m2048.md | cut -c1-150; grep "potrf n=\|relerr" $P/potrf2048.log
