#!/bin/bash
# kernel timeline of the headline step: busy/idle per step, largest gaps, per-kernel time (profiles/*_timeline_gaps.txt)
mkdir -p gpurun_out
R=$PWD
rm -rf $R/gpurun_out/prof_gap
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gap -o bench -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_gap.json 2> $R/gpurun_out/prof_gap.err
cd $R
python tools/gap_analysis.py $(find gpurun_out/prof_gap -name "*.db" | head -1) k_adam | tee gpurun_out/gap.txt | head -60
