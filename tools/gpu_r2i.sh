#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2i; mkdir -p $O; rm -f $O/summary.log
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "gemm or cfg4 or cfg5 or large_M or natgrad or round2" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -6 $O/pytest.log | cut -c1-300 >> $O/summary.log
for v in "DSDGP_GEMM_BIG=1" "DSDGP_GEMM_BIG=0"; do
  echo "== $v" >> $O/summary.log
  env $v timeout 300 python tools/bench_configs.py 4 5 2>&1 | cut -c1-200 >> $O/summary.log
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_cfg5 -o t -- python $R/tools/bench_configs.py 5 > $O/trace_cfg5.json 2> $O/trace_cfg5.err
db=$(find $O/trace_cfg5 -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $db $O/cfg5_kernel_stats.md "cfg5 mid-round" > /dev/null
head -24 $O/cfg5_kernel_stats.md | cut -c1-170 >> $O/summary.log
find $O -name "*.db" -size +30M -delete
cat $O/summary.log
