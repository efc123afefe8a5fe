#!/bin/bash
# SQ counters of the chain kernels of a config (default 3), serial schedule: where the waves' cycles go
cd "$GRAFT_REPO_ROOT" || exit 1
C=${1:-3}
R=$PWD; O=$R/gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  DSDGP_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -o pmc -- python $R/tools/ab_kernels.py $C > $O/$name.log 2> $O/$name.err
  DB=$(find $O/$name -name "*.db" | head -1)
  python $R/tools/pmc_table.py $DB k_layer k_wgrad > $O/$name.md
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE
{ echo "# SQ counters per launch shape, config $C (tools/ab_kernels.py $C), serial schedule (rocprofv3 --pmc, separate passes; averages per launch)"; echo;
  echo "## pass 1: cycles and waits"; echo; cat $O/sq1.md; echo; echo "## pass 2: instruction mix"; echo; cat $O/sq2.md; } > $O/pmc_sq_cfg$C.md
rm -rf $O/sq1 $O/sq2 $O/*.log $O/*.err; cat $O/pmc_sq_cfg$C.md | cut -c1-260
