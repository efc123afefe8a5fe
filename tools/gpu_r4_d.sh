#!/bin/bash
# LPT tile order of the grouped GEMM launches: full suite (default policy) + forced gemm path + config shapes + cfg 5 launch table
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; mkdir -p gpurun_out; O=$R/gpurun_out/r4d; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; tail -5 $O/t_all.log | cut -c1-250
DSDGP_FORCE=gemm_mp=16 timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q > $O/t_forced.log 2>&1; tail -5 $O/t_forced.log | cut -c1-250
timeout 600 python tools/bench_configs.py 1 2 3 4 5 2>&1 | grep "^{" | cut -c1-200
rm -rf /tmp/prof5
(cd /tmp && DSDGP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof5 -o p -- python $R/tools/ab_kernels.py 5 > $O/run5.log 2>&1)
DB=$(find /tmp/prof5 -name "*results.db" | head -1)
python $R/tools/launch_table.py $DB gemm_grouped pgemm > $O/launches5.md
grep "^{" $O/run5.log; cat $O/launches5.md | cut -c1-160
