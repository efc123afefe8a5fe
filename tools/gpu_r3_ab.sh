#!/bin/bash
# generic A/B of DSDGP_FORCE settings on config 2: AB="a|b|c" bash tools/gpu_r3_ab.sh
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r3ab; mkdir -p $O; export TMPDIR=/tmp
IFS='|' read -ra V <<< "$AB"
timeout 900 python tools/ab_force.py ${CFG:-2} "${V[@]}" > $O/ab.log 2> $O/ab.err
cat $O/ab.log
