#!/bin/bash
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
grep -o "SQ_[A-Z0-9_]*\|TCP_[A-Za-z0-9_]*\|TCC_[A-Za-z0-9_]*" $R/gpurun_out/counters_list.txt | sort -u > $R/gpurun_out/counter_names.txt
wc -l $R/gpurun_out/counter_names.txt
run() {
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmc_$name -o pmc -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_$name.json 2> $R/gpurun_out/pmc_$name.err
  echo "pmc $name exit $?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
tail -3 $R/gpurun_out/pmc_sq1.err $R/gpurun_out/pmc_sq2.err $R/gpurun_out/pmc_tcp.err | cut -c1-300
