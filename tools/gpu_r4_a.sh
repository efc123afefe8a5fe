#!/bin/bash
# round 4, first GPU batch: the GEMM-formulated large-M layers forced onto every parity shape, the full suite, A/B of the config shapes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; O=gpurun_out/r4a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== forced gemm path on the small parity shapes"
DSDGP_FORCE=gemm_mp=16 timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_parity.py -m gpu -q -x > $O/t_forced.log 2>&1; tail -15 $O/t_forced.log | cut -c1-300
echo "== full-size fixtures (default policy: gemm path from Mp = 512)"
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q > $O/t_full.log 2>&1; tail -25 $O/t_full.log | cut -c1-300
echo "== full suite"
timeout 1500 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; tail -8 $O/t_all.log | cut -c1-300
echo "== config shapes, default"
timeout 600 python tools/bench_configs.py 3 4 5 2>&1 | grep "^{" | cut -c1-200
echo "== config shapes, chains only (gemm_mp=0)"
DSDGP_FORCE=gemm_mp=0 timeout 600 python tools/bench_configs.py 4 5 2>&1 | grep "^{" | cut -c1-200
echo "== bench"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
