#!/bin/bash
# SQ counters of the config-2 training step (serial schedule: every launch alone on the chip): MFMA busy cycles, wave cycles, waits,
# instruction mix — evidence for DESIGN section 5.1's "the long phases run at the MFMA pipe's issue rate" (profiles/r03_pmc_sq.md)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/pmc_r3; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  DSDGP_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -o pmc -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/$name.json 2> $O/$name.err
  DB=$(find $O/$name -name "*.db" | head -1)
  python $R/tools/pmc_table.py $DB k_layer k_wgrad k_head k_reduce k_gemm_small k_asm_rows k_tail > $O/$name.md
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
{ echo "# round 3 — SQ counters per launch shape, config 2, serial schedule (rocprofv3 --pmc, separate passes; averages per launch)"; echo;
  echo "## pass 1: cycles and waits"; echo; cat $O/sq1.md; echo; echo "## pass 2: instruction mix"; echo; cat $O/sq2.md; } > $O/r03_pmc_sq.md
cat $O/r03_pmc_sq.md | cut -c1-200 | head -40
