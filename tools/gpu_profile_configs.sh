#!/bin/bash
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in ${CFGS:-3 4 5}; do
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg$c -o p -- python $R/tools/bench_configs.py $c > $R/gpurun_out/prof_cfg$c.json 2> $R/gpurun_out/prof_cfg$c.err
  echo "cfg$c exit $?"; cat $R/gpurun_out/prof_cfg$c.json
done
