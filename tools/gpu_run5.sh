#!/bin/bash
# new tests (full_cov) + PMC passes for HBM traffic of the dominant kernels
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "pytest exit $?" > gpurun_out/summary.log; tail -5 gpurun_out/t_all.log | cut -c1-300 >> gpurun_out/summary.log
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $R/gpurun_out/pmc_$ctr -o pmc -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_$ctr.json 2> $R/gpurun_out/pmc_$ctr.err
  echo "pmc $ctr exit $?" >> $R/gpurun_out/summary.log
done
cd $R
ls -R gpurun_out/pmc_FETCH_SIZE | head >> gpurun_out/summary.log
cat gpurun_out/summary.log
