#!/bin/bash
# round-2 GPU batch G: new Gram kernel (parity + GB/s), CS for Mp >= 512, kernel traces of cfg 4 / 5
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "gram or round2 or cfg4 or cfg5 or golden" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -4 $O/pytest.log | cut -c1-300 >> $O/summary.log
timeout 300 python tools/bench_configs.py 4 5 >> $O/ab.log 2>&1
timeout 300 python tools/ab_kernels.py 2 >> $O/ab.log 2>&1
grep -E "==|cfg|config" $O/ab.log | cut -c1-300 >> $O/summary.log
timeout 600 python bench.py --no-cpu-baseline --steps 50 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.log
python - <<PY >> $O/summary.log 2>&1
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], d["step_time"])
for g in d["sub_rooflines"]["gram"]: print(g)
PY
cd /tmp
for c in 5 4; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_cfg$c -o t -- python $R/tools/bench_configs.py $c > $O/trace_cfg$c.json 2> $O/trace_cfg$c.err
  db=$(find $O/trace_cfg$c -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py $db $O/cfg${c}_kernel_stats.md "round 2 (mid): config-$c shape, tools/bench_configs.py $c under rocprofv3 --kernel-trace --stats" > /dev/null
  head -40 $O/cfg${c}_kernel_stats.md | cut -c1-200 >> $O/summary.log
done
find $O -name "*.db" -size +30M -delete
cat $O/summary.log
