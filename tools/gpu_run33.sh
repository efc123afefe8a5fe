#!/bin/bash
mkdir -p gpurun_out
for t in 512 768 1024 1536 2048; do
  echo "target $t"; DSDGP_WGRAD_TARGET=$t timeout 300 python tools/ab_kernels.py 2 2>&1 | grep "^{"
done
for t in 768 1536; do
  echo "cfg3 target $t"; DSDGP_WGRAD_TARGET=$t timeout 300 python tools/ab_kernels.py 3 2>&1 | grep "^{"
done
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gap -o bench -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_gap.json 2> $R/gpurun_out/prof_gap.err
cd $R
python tools/gap_analysis.py $(find gpurun_out/prof_gap -name "*.db" | head -1) k_adam > gpurun_out/gap.txt; head -3 gpurun_out/gap.txt
