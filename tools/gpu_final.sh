#!/bin/bash
# round-end evidence: parity suite, smoke, full bench line (with cpu_baseline), rocprof kernel stats
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "pytest exit $?" > gpurun_out/summary.log; tail -3 gpurun_out/t_all.log | cut -c1-200 >> gpurun_out/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/summary.log; tail -1 gpurun_out/smoke.log >> gpurun_out/summary.log
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
echo "bench exit $?" >> gpurun_out/summary.log; cat gpurun_out/bench_final.json >> gpurun_out/summary.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
echo "rocprof exit $?" >> $R/gpurun_out/summary.log
cd $R
cat gpurun_out/summary.log
