#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dist_single.py -m gpu -q --no-header -p no:cacheprovider > gpurun_out/t_dist.log 2>&1
echo "pytest(dist) exit $?"; tail -15 gpurun_out/t_dist.log | cut -c1-300
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err
echo "torchrun bench exit $?"; cut -c1-300 gpurun_out/bench_torchrun1.json; tail -3 gpurun_out/bench_torchrun1.err
