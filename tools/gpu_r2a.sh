#!/bin/bash
# round-2 GPU batch A: parity of the alg_g / Csave paths + A/B of both on every config shape
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
for v in "DSDGP_SAVE_C=1 DSDGP_ALG_G=-1" "DSDGP_SAVE_C=0 DSDGP_ALG_G=-1" "DSDGP_SAVE_C=1 DSDGP_ALG_G=0" "DSDGP_SAVE_C=0 DSDGP_ALG_G=0"; do
  echo "== $v" >> $O/ab.log
  env $v timeout 300 python tools/ab_kernels.py 2 3 >> $O/ab.log 2>&1
  env $v timeout 300 python tools/bench_configs.py 4 5 >> $O/ab.log 2>&1
done
timeout 300 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
tail -1 $O/pytest.log | tee -a $O/summary.log
grep -E "==|cfg|config" $O/ab.log | cut -c1-260 | tee -a $O/summary.log
cut -c1-400 $O/bench.json | tee -a $O/summary.log
