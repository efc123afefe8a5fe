#!/bin/bash
# shapes beyond BASELINE.json (tools/bench_configs.py x6 .. x9): M > 1024, white = True at the config-4 / config-5 shapes (GEMM-formulated
# passes and, with gemm_mp=0, the chains); kernel stats and per-launch table of x6, dsdgp_potrf at n = 2048, dsdgp_trsm 1024 x 50000
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; P=$R/gpurun_out/large_m; rm -rf $P; mkdir -p $P
export TMPDIR=/tmp
timeout 600 python tools/bench_configs.py 6 7 8 9 > $P/r04_large_m_shapes.jsonl 2> $P/err.log
DSDGP_FORCE=gemm_mp=0 timeout 600 python tools/bench_configs.py 8 9 > $P/r04_white_shapes_chains_only.jsonl 2>> $P/err.log
rm -rf /tmp/prof6
(cd /tmp && DSDGP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof6 -o p -- python $R/tools/ab_kernels.py 6 > $P/run6.log 2>&1)
DB=$(find /tmp/prof6 -name "*results.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $P/r04_kernel_stats_m2048.md "round 4: 3-layer M = 2048 shape (tools/ab_kernels.py 6) under rocprofv3 --kernel-trace --stats, serial schedule" > /dev/null
[ -n "$DB" ] && python $R/tools/launch_table.py $DB pgemm kuf thin gl_ wgrad gemm_grouped chol > $P/r04_launch_shapes_m2048.md
timeout 200 python tools/potrf_prof.py 2048 > $P/potrf2048.log 2>&1
timeout 200 python tools/trsm_prof.py > $P/r04_trsm_1024x50000.txt 2>&1
cat $P/r04_large_m_shapes.jsonl $P/r04_white_shapes_chains_only.jsonl; grep "trsm trans" $P/r04_trsm_1024x50000.txt | tail -2; head -30 $P/r04_kernel_stats_m2048.md | cut -c1-150; grep "potrf n=\|relerr" $P/potrf2048.log
