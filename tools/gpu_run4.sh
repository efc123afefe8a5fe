#!/bin/bash
# parity suite on the default path + bench for chain-kernel variants + rocprof on the default
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/t_all.log 2>&1
echo "pytest(default) exit $?" > gpurun_out/summary.log; tail -3 gpurun_out/t_all.log >> gpurun_out/summary.log


for var in "DSDGP_CHAIN_SM=1"; do
  tag=$(echo "$var" | tr ' =' '__')
  env $var timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
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
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01c -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
echo "rocprof exit $?" >> $R/gpurun_out/summary.log
cd $R
cat gpurun_out/summary.log
