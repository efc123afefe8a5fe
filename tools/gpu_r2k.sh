#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2k; mkdir -p $O; rm -f $O/summary.log
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -4 $O/pytest.log | cut -c1-300 >> $O/summary.log
timeout 300 python tools/bench_configs.py 4 5 2>&1 | grep config | cut -c1-200 >> $O/summary.log
timeout 300 python tools/ab_kernels.py 2 2>&1 | grep cfg >> $O/summary.log
timeout 600 python bench.py --no-cpu-baseline --steps 50 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.log
python - <<PY >> $O/summary.log 2>&1
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], d["step_time"])
for g in d["sub_rooflines"]["potrf_trtri"]: print(g)
print(d["kernel_ms_per_step"])
PY
cat $O/summary.log
