#!/bin/bash
# round-2 GPU batch F: cooperative split-K wgrad; new bench.py protocol
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -4 $O/pytest.log | cut -c1-300 >> $O/summary.log
run() { echo "== $*" >> $O/ab.log; env "$@" timeout 300 python tools/ab_kernels.py 2 3 >> $O/ab.log 2>&1; }
run DSDGP_WGRAD_COOP=1
run DSDGP_WGRAD_COOP=1 DSDGP_WGRAD_TARGET=768
run DSDGP_WGRAD_COOP=1 DSDGP_WGRAD_TARGET=1024
run DSDGP_WGRAD_COOP=1 DSDGP_WGRAD_TARGET=384
run DSDGP_WGRAD_COOP=0
run DSDGP_WGRAD_COOP=0 DSDGP_WGRAD_TARGET=4096
for v in "DSDGP_WGRAD_COOP=1" "DSDGP_WGRAD_COOP=1 DSDGP_WGRAD_TARGET=1024" "DSDGP_WGRAD_COOP=0 DSDGP_WGRAD_TARGET=4096"; do
  echo "== $v" >> $O/ab.log
  env $v timeout 300 python tools/bench_configs.py 1 4 5 >> $O/ab.log 2>&1
done
grep -E "==|cfg|config" $O/ab.log | cut -c1-300 >> $O/summary.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.log
tail -3 $O/bench.err >> $O/summary.log
python - <<PY >> $O/summary.log 2>&1
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","step_time","roofline","sub_rooflines","cpu_baseline","elbo_evals_per_s","predict_f_rows_per_s"):
    print(k, json.dumps(d.get(k)))
PY
cat $O/summary.log
