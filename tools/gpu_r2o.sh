#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2o; mkdir -p $O; rm -f $O/summary.log
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.log; tail -1 $O/smoke.log >> $O/summary.log
for sc in strong weak; do
DSDGP_BENCH_BACKEND=gloo DSDGP_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --scaling $sc > $O/bench_n2_$sc.json 2> $O/bench_n2_$sc.err
echo "bench n2 $sc rc=$?" >> $O/summary.log
tail -1 $O/bench_n2_$sc.json | cut -c1-700 >> $O/summary.log
tail -3 $O/bench_n2_$sc.err | cut -c1-300 >> $O/summary.log
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench torchrun n1 rc=$?" >> $O/summary.log
tail -1 $O/bench_n1.json | cut -c1-400 >> $O/summary.log
cat $O/summary.log
