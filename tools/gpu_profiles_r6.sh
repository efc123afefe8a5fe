#!/bin/bash
# round-6 evidence batch: parity suite, PMC traffic of the headline config (with the hash of the kernel sources) and of configs 3 / 5,
# bench line, kernel stats (overlap-free and production) + timeline of config 2, kernel stats and per-launch tables of configs 3 / 4 / 5
# (GEMM-formulated large-M layers), all config shapes, potrf at n = 1024
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/prof_r6; rm -rf $O; mkdir -p $O
P=$R/gpurun_out/profiles_r6; rm -rf $P; mkdir -p $P
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $P/pytest.log 2>&1; echo "pytest rc=$?" > $P/summary.log; grep "passed\|failed" $P/pytest.log >> $P/summary.log
H=$(python -c "import bench; print(bench.csrc_hash())")
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  DSDGP_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_$ctr -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_$ctr.json 2> $O/pmc_$ctr.err
done
python $R/tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) $P/r06_pmc_traffic "round 6 — HBM traffic per launch (rocprofv3 --pmc, cfg 2)" $H
cp $P/r06_pmc_traffic.json $R/profiles/r06_pmc_traffic.json
# config 3 / config 5 traffic (tools/ab_kernels.py: a handful of training steps of that shape)
for c in 3 5; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    DSDGP_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc${c}_$ctr -o p -- python $R/tools/ab_kernels.py $c > $O/pmc${c}_$ctr.log 2>&1
  done
  python $R/tools/pmc_traffic.py $(find $O/pmc${c}_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc${c}_WRITE_SIZE -name "*.db" | head -1) $P/r06_pmc_traffic_cfg$c "round 6 — HBM traffic per launch (rocprofv3 --pmc), config-$c shape (tools/ab_kernels.py $c)" $H
done
# executed fp64 MFMA flops per step of every config shape (SQ_INSTS_VALU_MFMA_MOPS_F64 x 512): bench.py's executed_gflop_per_step
EX=""
for c in 1 2 3 4 5; do
  n=$([ $c -le 2 ] && echo 20 || echo 4)
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $O/ex$c -o p -- python $R/tools/executed_flops.py run $c $n > $O/ex$c.log 2>&1
  DB=$(find $O/ex$c -name "*.db" | head -1)
  [ -n "$DB" ] && EX="$EX cfg$c=$DB:$n"
done
python $R/tools/executed_flops.py parse $P/r06_executed_flops.json $H $EX >> $P/summary.log 2>&1
cp $P/r06_executed_flops.json $R/profiles/r06_executed_flops.json
(cd $R && timeout 900 python bench.py > $P/r06_bench.json 2> $O/bench.err; tail -c 400 $P/r06_bench.json >> $P/summary.log)
DSDGP_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/serial -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/serial.json 2> $O/serial.err
DB=$(find $O/serial -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $P/r06_kernel_stats_serial.md "round 6: bench.py --steps 40 --warmup 5 --no-extras under rocprofv3 --kernel-trace --stats, DSDGP_NO_OVERLAP=1 (overlap-free: every duration is the kernel's own)" > /dev/null
python $R/tools/launch_table.py $DB layer_ wgrad gemm head reduce asm tail > $P/r06_launch_shapes_serial.md
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prod -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/prod.json 2> $O/prod.err
DB=$(find $O/prod -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $P/r06_kernel_stats.md "round 6: bench.py --steps 40 --warmup 5 --no-extras under rocprofv3 --kernel-trace --stats (production: side-stream overlap on, durations of co-running kernels stretch)" > /dev/null
python $R/tools/gap_analysis.py $DB k_tail > $P/r06_timeline_gaps.txt
python $R/tools/timeline_dump.py $DB k_tail 3 > $P/r06_timeline_step.txt
cd $R
timeout 900 python tools/bench_configs.py 1 2 3 4 5 > $P/r06_all_config_shapes.jsonl 2> $O/all.err
DSDGP_FORCE=gemm_mp=0 timeout 600 python tools/bench_configs.py 4 5 > $P/r06_config_shapes_chains_only.jsonl 2> $O/all0.err
timeout 300 python tools/gemm_bench.py > $P/r06_gemm_bench.txt 2> $O/gemm.err
for c in 3 4 5; do
  rm -rf /tmp/prof$c
  (cd /tmp && DSDGP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$c -o p -- python $R/tools/ab_kernels.py $c > $O/run$c.log 2>&1)
  DB=$(find /tmp/prof$c -name "*results.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $P/r06_kernel_stats_cfg$c.md "round 6: config-$c shape (tools/ab_kernels.py $c) under rocprofv3 --kernel-trace --stats, serial schedule" > /dev/null
  [ -n "$DB" ] && python $R/tools/launch_table.py $DB pgemm kuf thin gl_ layer_ wgrad gemm_grouped chol > $P/r06_launch_shapes_cfg$c.md
  grep "^{" $O/run$c.log >> $P/r06_ab_kernels.txt
done
rm -rf /tmp/pg; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pg -o p -- python $R/tools/gram_time.py > $P/r06_gram_time.txt 2>&1)
DB=$(find /tmp/pg -name "*results.db" | head -1)
[ -n "$DB" ] && python $R/tools/launch_table.py $DB gram > $P/r06_gram_launches.md
rm -rf /tmp/pp; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python $R/tools/potrf_prof.py 1024 > $O/potrf.log 2>&1)
DB=$(find /tmp/pp -name "*results.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $P/r06_potrf_n1024_stats.md "round 6: dsdgp_potrf n = 1024, 6 calls, then torch.linalg.cholesky (rocSOLVER) of the same matrix once" > /dev/null
grep "potrf n=\|relerr" $O/potrf.log > $P/r06_potrf_n1024_wall.txt
# factor + inverse of Ku through the model path (what sub_rooflines.potrf_trtri times): look-ahead sequence against the plain blocked one, then
# the launch timeline of one M = 1024 sequence and dsdgp_potrf's orders / residuals
(cd $R && (echo "== look-ahead sequence (default)"; timeout 300 python tools/potrf_model_time.py 192 256 512 1024 2048; echo "== DSDGP_CHOL_LOOKAHEAD=0 (plain blocked sequence, recursive-doubling inverse)"; DSDGP_CHOL_LOOKAHEAD=0 timeout 300 python tools/potrf_model_time.py 192 256 512 1024 2048; echo "== dsdgp_potrf (wall, pad + factor + unpad)"; timeout 300 python tools/potrf_check.py) 2>&1 | grep -v amdgpu.ids > $P/r06_potrf_model_path.txt)
rm -rf /tmp/pm; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pm -o p -- python $R/tools/potrf_model_time.py 1024 > /dev/null 2>&1)
DB=$(find /tmp/pm -name "*results.db" | head -1)
[ -n "$DB" ] && python $R/tools/timeline_dump.py $DB k_chol_xrow 3 > $P/r06_potrf_model_path_timeline.txt
# shard steps of a strong-scaling run (1000 / 500 / 250 / 125 rows: fused, elbo + adam, data-parallel flat / bucketed over a one-rank RCCL group)
(cd $R && timeout 600 python tools/bench_shards.py 2>/dev/null | grep "^{" > $P/r06_strong_scaling_shards.jsonl)
# one step's timeline at the 125-row shard
rm -rf /tmp/tl125; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/tl125 -o t -- python $R/tools/shard_timeline.py 125 > $O/tl125.log 2>&1)
DB=$(find /tmp/tl125 -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/timeline_dump.py $DB k_tail 3 > $P/r06_timeline_step_125rows.txt
rm -rf $O
cat $P/summary.log; cat $P/r06_all_config_shapes.jsonl | cut -c1-160; cat $P/r06_config_shapes_chains_only.jsonl | cut -c1-160
