#!/bin/bash
# round 5, batch A: parity suite on the new build, event-flag A/B, Cholesky worker-selection A/B, the new bench line
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r5a; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/summary.log; grep "passed\|failed" $O/pytest.log >> $O/summary.log
DSDGP_CHOL_SCHED=1 timeout 600 python -m pytest tests -m gpu -q -x -k "potrf or chol or golden or full_size or natgrad or large_M" > $O/pytest_sched1.log 2>&1; echo "pytest sched1 rc=$?" >> $O/summary.log; grep "passed\|failed" $O/pytest_sched1.log >> $O/summary.log
echo "== ev_fence A/B cfg2" >> $O/summary.log
timeout 300 python tools/ab_force.py 2 "ev_fence=0" "ev_fence=1" "bwd_split=2" 2>&1 | grep "^{" >> $O/summary.log
echo "== ev_fence A/B cfg5" >> $O/summary.log
timeout 300 python tools/ab_force.py 5 "ev_fence=0" "ev_fence=1" 2>&1 | grep "^{" >> $O/summary.log
for s in 0 1 0 1; do
  echo "== chol sched $s" >> $O/summary.log
  DSDGP_CHOL_SCHED=$s timeout 300 python tools/ab_kernels.py 2 5 2>&1 | grep "^{" >> $O/summary.log
done
for s in 0 1; do
  DSDGP_CHOL_SCHED=$s DSDGP_POTRF_TIMING=1 timeout 120 python tools/potrf_timing.py 2>&1 | grep "head cycles" | tail -2 >> $O/summary.log
  DSDGP_CHOL_SCHED=$s timeout 120 python tools/potrf_prof.py 1024 2>&1 | tail -3 >> $O/summary.log
done
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.log
python - <<PY >> $O/summary.log
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "launches", d.get("launches_per_step"), "steady", d["step_time"])
print("roofline", {k:d["roofline"][k] for k in ("kernel","frac","frac_executed","ms_per_step")})
for c in d["all_configs"]: print(c["config"][:5], c["steps_per_s"], c["step_time"], c["launches_per_step"], c["frac_of_fp64_peak"])
print(d["kernel_ms_per_step"]["potrf"], d["sub_rooflines"]["potrf_trtri"])
PY
cat $O/summary.log
