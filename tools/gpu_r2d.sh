#!/bin/bash
# round-2 GPU batch D: light/heavy-wave Csave chain
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -15 $O/pytest.log | cut -c1-300 >> $O/summary.log
run() { echo "== $*" >> $O/ab.log; env "$@" timeout 300 python tools/ab_kernels.py 2 3 >> $O/ab.log 2>&1; }
run DSDGP_SAVE_C=1
run DSDGP_SAVE_C=0
run DSDGP_SAVE_C=1 DSDGP_WGRAD_TARGET=4096
run DSDGP_SAVE_C=0 DSDGP_WGRAD_TARGET=4096
run DSDGP_SAVE_C=1 DSDGP_CS_MIN_DOUT=1
grep -E "==|cfg|config" $O/ab.log | cut -c1-260 >> $O/summary.log
cd /tmp
for sc in 1; do
  DSDGP_NO_OVERLAP=1 DSDGP_SAVE_C=$sc timeout 300 rocprofv3 --kernel-trace -d $O/trace_c$sc -o t -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/trace_c$sc.json 2> $O/trace_c$sc.err
  echo "== trace SAVE_C=$sc" >> $O/summary.log
  python $R/tools/launch_table.py $(find $O/trace_c$sc -name "*.db" | head -1) layer_ wgrad >> $O/summary.log
  DSDGP_NO_OVERLAP=1 DSDGP_SAVE_C=$sc timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_c$sc -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_c$sc.json 2> $O/pmc_c$sc.err
  python $R/tools/pmc_table.py $(find $O/pmc_c$sc -name "*.db" | head -1) layer_bwd layer_fwd >> $O/summary.log 2>&1
done
find $O -name "*.db" -size +30M -delete
cat $O/summary.log
