#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python tools/bench_configs.py > gpurun_out/bench_configs.jsonl 2> gpurun_out/bench_configs.err
echo "exit $?"; cat gpurun_out/bench_configs.jsonl; tail -5 gpurun_out/bench_configs.err
