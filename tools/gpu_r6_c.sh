#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6c; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 60 tools/bin/anyorder_probe > $O/anyorder.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_round6.py -m gpu -q -x > $O/t_r6.log 2>&1; echo "r6 pytest rc=$?" >> $O/summary.log; tail -5 $O/t_r6.log >> $O/summary.log
cat $O/anyorder.txt $O/summary.log
