#!/bin/bash
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/t_all.log 2>&1
echo "pytest exit $?" > gpurun_out/summary.log; tail -3 gpurun_out/t_all.log | cut -c1-200 >> gpurun_out/summary.log
for var in "DSDGP_NO_OVERLAP=0" "DSDGP_NO_OVERLAP=1"; do
  tag=$(echo "$var" | tr ' =' '__')
  env $var timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
  echo "bench [$var] exit $?" >> gpurun_out/summary.log
  python - <<PY >> gpurun_out/summary.log
import json
try:
    d=json.load(open("gpurun_out/bench_$tag.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in d["kernel_ms_per_step"].items()}, "evals/s", d["elbo_evals_per_s"], "pred rows/s", d["predict_f_rows_per_s"])
except Exception as e:
    print("bench parse fail", e); print(open("gpurun_out/bench_$tag.err").read()[-2000:])
PY
done
cat gpurun_out/summary.log
