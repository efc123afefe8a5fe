#!/bin/bash
# large-M check: tests, config shapes 3 / 4 / 5, potrf + trsm sub-rooflines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3big; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
for c in ${CFGS:-3 4 5}; do timeout 300 python tools/ab_kernels.py $c 2>&1 | grep "^{"; done
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | grep "^{" > $O/bench.json
python - <<'P'
import json
b = json.load(open("gpurun_out/r3big/bench.json"))
print(b["value"], b["ms_per_step"])
for k in ("potrf_trtri", "trsm"):
    for e in b["sub_rooflines"][k]: print(k, {x: e[x] for x in e if x in ("n", "nrhs", "us", "frac")})
P
