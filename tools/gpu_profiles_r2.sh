#!/bin/bash
# round-2 evidence: kernel stats (overlap-free and production), timeline gaps, PMC traffic, SQ stall counters, Gram HBM counters,
# Cholesky / GEMM MFMA utilisation at M = 1024, all config shapes, padding cost, fp64 MFMA clock + power samples
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/prof_r2; rm -rf $O; mkdir -p $O
P=$R/gpurun_out/profiles_r2; rm -rf $P; mkdir -p $P
export TMPDIR=/tmp
cd /tmp
# 1. headline step, overlap-free (each duration is the kernel's own) and production (side-stream overlap on)
DSDGP_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/serial -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/serial.json 2> $O/serial.err
python $R/tools/rocprof_summary.py $(find $O/serial -name "*.db" | head -1) $P/r02_kernel_stats_serial.md "round 2: bench.py --steps 40 --warmup 5 --no-extras under rocprofv3 --kernel-trace --stats, DSDGP_NO_OVERLAP=1 (overlap-free: every duration is the kernel's own)" > /dev/null
python $R/tools/launch_table.py $(find $O/serial -name "*.db" | head -1) layer_ wgrad gemm potrf reduce > $P/r02_launch_shapes_serial.md
python $R/tools/gap_analysis.py $(find $O/serial -name "*.db" | head -1) k_adam > $P/r02_timeline_gaps.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prod -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/prod.json 2> $O/prod.err
python $R/tools/rocprof_summary.py $(find $O/prod -name "*.db" | head -1) $P/r02_kernel_stats.md "round 2: bench.py --steps 40 --warmup 5 --no-extras under rocprofv3 --kernel-trace --stats (production: side-stream overlap on, durations of co-running kernels stretch)" > /dev/null
# 2. HBM traffic (separate passes)
for ctr in FETCH_SIZE WRITE_SIZE; do
  DSDGP_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_$ctr -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_$ctr.json 2> $O/pmc_$ctr.err
done
python $R/tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) $P/r02_pmc_traffic "round 2 — HBM traffic per launch (rocprofv3 --pmc, cfg 2)"
# 3. SQ stall counters
DSDGP_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_sq -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_sq.json 2> $O/pmc_sq.err
python $R/tools/pmc_table.py $(find $O/pmc_sq -name "*.db" | head -1) layer_ wgrad gemm potrf > $P/r02_pmc_sq_stalls.md
# 4. Gram sub-roofline: counters of the dsdgp_gram launches
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $O/gram_$ctr -o p -- python $R/tools/gram_pmc.py > $O/gram_$ctr.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/gram_t -o t -- python $R/tools/gram_pmc.py > $O/gram_t.log 2>&1
python $R/tools/launch_table.py $(find $O/gram_t -name "*.db" | head -1) gram > $P/r02_gram_launches.md
python $R/tools/pmc_table.py $(find $O/gram_FETCH_SIZE -name "*.db" | head -1) gram > $P/r02_gram_fetch.md
python $R/tools/pmc_table.py $(find $O/gram_WRITE_SIZE -name "*.db" | head -1) gram > $P/r02_gram_write.md
# 5. config 5 / 4 / 3: kernel stats + MFMA utilisation of the Cholesky / GEMM launches at M = 1024
for c in 3 4 5; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/cfg$c -o t -- python $R/tools/bench_configs.py $c > $O/cfg$c.json 2> $O/cfg$c.err
  python $R/tools/rocprof_summary.py $(find $O/cfg$c -name "*.db" | head -1) $P/r02_cfg${c}_kernel_stats.md "round 2: config-$c shape, tools/bench_configs.py $c under rocprofv3 --kernel-trace --stats" > /dev/null
done
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $O/cfg5_pmc -o p -- python $R/tools/bench_configs.py 5 > $O/cfg5_pmc.json 2> $O/cfg5_pmc.err
python $R/tools/pmc_table.py $(find $O/cfg5_pmc -name "*.db" | head -1) potrf gemm trtri > $P/r02_cfg5_cholesky_gemm_pmc.md
cd $R
# 6. all config shapes, padding cost
timeout 600 python tools/bench_configs.py 1 2 3 4 5 > $P/r02_all_config_shapes.jsonl 2> $O/all.err
timeout 600 python tools/bench_padding.py > $P/r02_padding_cost.jsonl 2> $O/pad.err
# 7. fp64 MFMA clock microbenchmark; then ~8 s of the saturating loop with power / clock samples taken DURING it
timeout 120 tools/bin/mfma_clock_bench > $P/r02_mfma_clock_microbench.txt 2>&1
( timeout 60 tools/bin/mfma_clock_bench 8 > $O/mfma_long.txt 2>&1 ) &
LONG=$!
sleep 1.0
( while kill -0 $LONG 2>/dev/null; do /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; done ) > $O/smi_during.jsonl
wait $LONG
python - <<PY > $P/r02_mfma_power_samples.txt 2>&1
import json
rows=[]
for line in open("$O/smi_during.jsonl"):
    try:
        d=json.loads(line)
    except Exception:
        continue
    for card,v in d.items():
        if isinstance(v,dict):
            rows.append({k:v[k] for k in v if any(s in k.lower() for s in ("power","sclk"))})
print("rocm-smi --showpower --showclocks samples taken back to back WHILE tools/bin/mfma_clock_bench ran its saturating")
print("configuration (v_mfma_f64_16x16x4_f64, 8 accumulators, 2 waves / SIMD, ~47 TFLOP/s) for ~8 s:")
for r in rows: print(r)
tail = open("$O/mfma_long.txt").read().strip().splitlines()[-3:]
print("last lines of the loop's own report:")
for t in tail: print(t)
PY
# 7b. microbenchmarks of the fp64 MFMA forms and of the chain loop's ingredients; per-phase clocks of the chain kernels
for t in mfma_f64_variants mfma_4x4_probe mfma_emul_test mfma_clock_probe chain_loop_bench; do
  timeout 120 tools/bin/$t > $P/r02_$t.txt 2>&1
done
timeout 300 python tools/bwd_phases.py 2 3 4 5 2> $O/phases.txt > /dev/null
( echo "# DSDGP_FWD_TIMING=1 DSDGP_BWD_TIMING=1 DSDGP_NO_OVERLAP=1 python tools/bwd_phases.py 2 3 4 5  (third step of each config;"; echo "# shader clocks per workgroup averaged over the launch)"; grep -E "^== cfg|phases\]" $O/phases.txt | awk '/^== cfg/{c=$0; n=0; print; next} {print}' | sed 's/   launch span.*//' ) > $P/r02_chain_phases.txt
# 8. headline bench line (full, with cpu_baseline)
timeout 900 python bench.py > $P/r02_bench.json 2> $O/bench.err
find $O -name "*.db" -size +20M -delete
ls -la $P
tail -c 1500 $P/r02_bench.json
