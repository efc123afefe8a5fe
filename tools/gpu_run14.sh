#!/bin/bash
mkdir -p gpurun_out
( for i in 1 2 3 4 5 6 7 8; do sleep 0.4; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Socket" ; done ) > gpurun_out/clocks.txt 2>&1 &
CPID=$!
./tools/bin/mfma_f64_bench | tee gpurun_out/mfma_bench.txt
wait $CPID
cat gpurun_out/clocks.txt | head -30
