#!/bin/bash
# quick A/B: config-2 step time + per-kernel event times, twice per library in LIBS (default = in-tree build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3q; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/ab.log
for rnd in 1 2; do for lib in ${LIBS:-default}; do
  if [ "$lib" = "default" ]; then unset DSDGP_LIB_PATH; else export DSDGP_LIB_PATH=$PWD/$lib; fi
  echo "== $lib ${FORCE}" >> $O/ab.log
  DSDGP_FORCE="$FORCE" timeout 200 python tools/ab_kernels.py ${CFG:-2} 2>&1 | grep "^{" >> $O/ab.log
done; done
cat $O/ab.log
