#!/bin/bash
# round-3 evidence batch: parity suite, bench line, kernel stats (overlap-free and production), one step's timeline + gaps, PMC traffic
# (with the hash of the kernel sources), chain / head phase clocks, all config shapes, strong-scaling shard sizes, pivot-loop microbench
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/prof_r3; rm -rf $O; mkdir -p $O
P=$R/gpurun_out/profiles_r3; rm -rf $P; mkdir -p $P
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $P/pytest.log 2>&1; echo "pytest rc=$?" > $P/summary.log; tail -3 $P/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" >> $P/summary.log
H=$(python -c "import bench; print(bench.csrc_hash())")
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  DSDGP_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_$ctr -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_$ctr.json 2> $O/pmc_$ctr.err
done
python $R/tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) $P/r03_pmc_traffic "round 3 — HBM traffic per launch (rocprofv3 --pmc, cfg 2)" $H
# the bench line AFTER the counters of this build are in place (bench.py reports traffic only when the source hashes match)
cp $P/r03_pmc_traffic.json $R/profiles/r03_pmc_traffic.json
(cd $R && timeout 900 python bench.py > $P/r03_bench.json 2> $O/bench.err; tail -c 600 $P/r03_bench.json >> $P/summary.log)
DSDGP_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/serial -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/serial.json 2> $O/serial.err
DB=$(find $O/serial -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $P/r03_kernel_stats_serial.md "round 3: bench.py --steps 40 --warmup 5 --no-extras under rocprofv3 --kernel-trace --stats, DSDGP_NO_OVERLAP=1 (overlap-free: every duration is the kernel's own)" > /dev/null
python $R/tools/launch_table.py $DB layer_ wgrad gemm head reduce asm tail > $P/r03_launch_shapes_serial.md
python $R/tools/gap_analysis.py $DB k_tail > $P/r03_timeline_gaps_serial.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prod -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/prod.json 2> $O/prod.err
DB=$(find $O/prod -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $P/r03_kernel_stats.md "round 3: bench.py --steps 40 --warmup 5 --no-extras under rocprofv3 --kernel-trace --stats (production: side-stream overlap on, durations of co-running kernels stretch)" > /dev/null
python $R/tools/gap_analysis.py $DB k_tail > $P/r03_timeline_gaps.txt
python $R/tools/timeline_dump.py $DB k_tail 3 > $P/r03_timeline_step.txt
cd $R
DSDGP_FWD_TIMING=1 DSDGP_BWD_TIMING=1 DSDGP_NO_OVERLAP=1 timeout 600 python tools/bwd_phases.py 2 3 > $P/r03_chain_phases.txt 2>&1
DSDGP_POTRF_TIMING=1 timeout 300 python tools/potrf_timing.py 2>&1 | grep cycles > $P/r03_head_phases.txt
timeout 900 python tools/bench_configs.py 1 2 3 4 5 > $P/r03_all_config_shapes.jsonl 2> $O/all.err
timeout 600 python tools/bench_shards.py > $P/r03_strong_scaling_shards.jsonl 2> $O/shards.err
timeout 120 tools/bin/chol16_bench > $P/r03_chol16_bench.txt 2>&1
timeout 300 python tools/gemm_bench.py > $P/r03_gemm_bench.txt 2> $O/gemm.err
# large-M shapes: kernel stats of configs 4 / 5 (serial schedule) and of dsdgp_potrf at n = 1024
for c in 4 5; do
  rm -rf /tmp/prof$c
  (cd /tmp && DSDGP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$c -o p -- python $R/tools/ab_kernels.py $c > $O/run$c.log 2>&1)
  DB=$(find /tmp/prof$c -name "*results.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $P/r03_kernel_stats_cfg$c.md "round 3: config-$c shape (tools/ab_kernels.py $c) under rocprofv3 --kernel-trace --stats, serial schedule" > /dev/null
done
rm -rf /tmp/pp; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python $R/tools/potrf_prof.py 1024 > $O/potrf.log 2>&1)
DB=$(find /tmp/pp -name "*results.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $P/r03_potrf_n1024_stats.md "round 3: dsdgp_potrf n = 1024, 6 calls, then torch.linalg.cholesky (rocSOLVER) of the same matrix once" > /dev/null
grep "potrf n=\|relerr" $O/potrf.log > $P/r03_potrf_n1024_wall.txt
rm -rf $O
cat $P/summary.log; cat $P/r03_all_config_shapes.jsonl | cut -c1-160; cat $P/r03_strong_scaling_shards.jsonl
