#!/bin/bash
# A/B of library builds on config 2: LIBS="a.so b.so" (paths relative to the repository root; "default" = the in-tree build)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r3lib; mkdir -p $O; export TMPDIR=/tmp
rm -f $O/ab.log
for rnd in 1 2; do
for lib in $LIBS; do
  if [ "$lib" = "default" ]; then unset DSDGP_LIB_PATH; else export DSDGP_LIB_PATH=$R/$lib; fi
  echo "== $lib" >> $O/ab.log
  timeout 300 python tools/ab_kernels.py ${CFG:-2} 2>&1 | grep "^{" >> $O/ab.log
done
done
cat $O/ab.log
