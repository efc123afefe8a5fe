#!/bin/bash
# round-end evidence: parity suite, smoke, full bench line (with cpu_baseline), rocprof kernel stats, PMC traffic, all config shapes
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "pytest exit $?" > gpurun_out/summary.log; grep "passed\|failed" gpurun_out/t_all.log >> gpurun_out/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/summary.log; tail -1 gpurun_out/smoke.log >> gpurun_out/summary.log
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
echo "bench exit $?" >> gpurun_out/summary.log; cat gpurun_out/bench_final.json >> gpurun_out/summary.log
timeout 900 python tools/bench_configs.py 1 2 3 4 5 > gpurun_out/all_configs.jsonl 2> gpurun_out/all_configs.err
echo "configs exit $?" >> gpurun_out/summary.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
echo "rocprof exit $?" >> $R/gpurun_out/summary.log
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $R/gpurun_out/pmc_$ctr -o pmc -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_$ctr.json 2> $R/gpurun_out/pmc_$ctr.err
  echo "pmc $ctr exit $?" >> $R/gpurun_out/summary.log
done
cd $R
cat gpurun_out/summary.log
