#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2n; mkdir -p $O; rm -f $O/summary.log
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -4 $O/pytest.log | cut -c1-300 >> $O/summary.log
timeout 300 python tools/ab_kernels.py 2 3 2>&1 | grep cfg >> $O/summary.log
cd /tmp
DSDGP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/trace.json 2> $O/trace.err
python $R/tools/launch_table.py $(find $O/trace -name "*.db" | head -1) layer_ >> $O/summary.log
find $O -name "*.db" -delete
cat $O/summary.log
