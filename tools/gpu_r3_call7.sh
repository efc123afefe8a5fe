#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c7; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/summary.txt
tail -3 $O/pytest.log >> $O/summary.txt
for i in 1 2; do timeout 200 python tools/ab_kernels.py 2 2>&1 | grep "^{" >> $O/summary.txt; done
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" > $O/bench.json
python - <<'P' >> $O/summary.txt
import json
b=json.load(open("gpurun_out/r3c7/bench.json")); print(b["value"], b["ms_per_step"], b.get("kernel_ms"))
P
cat $O/summary.txt
