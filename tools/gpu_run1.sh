#!/bin/bash
# first GPU bring-up: fp64 MFMA rate + parity tests by group (each in its own process)
mkdir -p gpurun_out
tools/bin/mfma_f64_bench > gpurun_out/mfma.log 2>&1
for grp in "gemm or potrf or trsm or gram or randn" "layer_conditional or cholesky_failure" "propagate or elbo" "gradients" "adam or trainable or mc_elbo or predict or minibatch"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "$grp" > "gpurun_out/t_${name}.log" 2>&1
  echo "== $grp : exit $?" >> gpurun_out/summary.log
  tail -3 "gpurun_out/t_${name}.log" >> gpurun_out/summary.log
done
cat gpurun_out/mfma.log gpurun_out/summary.log
