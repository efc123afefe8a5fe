#!/bin/bash
# round-4 regression check of the large-M path: full-size fixtures, the GEMM-formulated layers forced onto every parity shape
# (DSDGP_FORCE=gemm_mp=16), config shapes 4 / 5, rocprofv3 kernel stats + per-launch tables of both
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; mkdir -p gpurun_out; O=$R/gpurun_out/r4e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q > $O/t_full.log 2>&1; tail -5 $O/t_full.log | cut -c1-250
DSDGP_FORCE=gemm_mp=16 timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q > $O/t_forced.log 2>&1; tail -5 $O/t_forced.log | cut -c1-250
timeout 600 python tools/bench_configs.py 4 5 2>&1 | grep "^{" | cut -c1-200
for c in 4 5; do
rm -rf /tmp/prof$c
(cd /tmp && DSDGP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$c -o p -- python $R/tools/ab_kernels.py $c > $O/run$c.log 2>&1)
DB=$(find /tmp/prof$c -name "*results.db" | head -1)
python $R/tools/launch_table.py $DB pgemm kuf thin gl_ > $O/launches$c.md
python $R/tools/rocprof_summary.py $DB $O/kernel_stats_cfg$c.md "round 4: config-$c shape (tools/ab_kernels.py $c) under rocprofv3 --kernel-trace --stats, serial schedule" > /dev/null
grep "^{" $O/run$c.log; cat $O/launches$c.md | cut -c1-160
done
