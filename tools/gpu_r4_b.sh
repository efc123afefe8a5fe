#!/bin/bash
# round 4: forced GEMM path on the parity shapes + rocprofv3 kernel stats of configs 4 / 5 (serial schedule)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; mkdir -p gpurun_out; O=$R/gpurun_out/r4b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
DSDGP_FORCE=gemm_mp=16,wgrad_mp=16 timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q > $O/t_forced.log 2>&1; tail -12 $O/t_forced.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q > $O/t_full.log 2>&1; tail -12 $O/t_full.log | cut -c1-250
timeout 600 python tools/bench_configs.py 3 4 5 2>&1 | grep "^{" | cut -c1-200
echo "== A/B: operand-streaming weight-gradient products everywhere (wgrad_mp=0)"
DSDGP_FORCE=wgrad_mp=0 timeout 600 python tools/bench_configs.py 3 4 5 2>&1 | grep "^{" | cut -c1-200
echo "== A/B: GEMM weight-gradient products from Mp = 128 (cfg 2)"
timeout 600 python tools/bench_configs.py 2 2>&1 | grep "^{" | cut -c1-200
DSDGP_FORCE=wgrad_mp=128 timeout 600 python tools/bench_configs.py 2 2>&1 | grep "^{" | cut -c1-200
for c in 3 5; do
  rm -rf /tmp/prof$c
  (cd /tmp && DSDGP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$c -o p -- python $R/tools/ab_kernels.py $c > $O/run$c.log 2>&1)
  DB=$(find /tmp/prof$c -name "*results.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $O/kernel_stats_cfg$c.md "round 4 (gemm path, fourth version: GEMM weight-gradient products): config-$c shape (tools/ab_kernels.py $c) under rocprofv3 --kernel-trace --stats, serial schedule" > /dev/null
  grep "^{" $O/run$c.log
  head -22 $O/kernel_stats_cfg$c.md | cut -c1-200
done
