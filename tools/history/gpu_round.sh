#!/bin/bash
# One GPU call with everything a round's profiles/ need: tests (+ the long thresholding sweep), bench lines (fp16 / fp32 /
# bf16 / fp32+fp16), rocprofv3 kernel trace (+ kernel_stats.csv) and memory-side counters of the headline command, the stage
# kernel inside the torch network loop (events + rocprofv3 rows), the stage table.   usage (through gpurun): bash tools/gpu_round.sh r03
TAG=${1:-r03}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
[ -f _refscratch/dpm_solver_pytorch.py ] && export DPM_REFERENCE_DIR=_refscratch
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
( time DPM_THR_SWEEP=20000 DPM_THR_SWEEP_STEPS=20 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_sweep" ) > $O/pytest_thr_sweep_20000.log 2>&1; echo "thresholding sweep (20000 configurations, up to 19 steps) rc=$?"; grep -E "passed|failed|real" $O/pytest_thr_sweep_20000.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
# the launch line the driver uses for N > 1, with one rank (RCCL communicator, barrier, MAX all-reduce, final all-gather)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun_1rank.err; echo "torchrun 1 rank rc=$?"
timeout 900 bash tools/profile_round.sh $TAG fp16 > $O/profile_fp16.log 2>&1; echo "profile rc=$?"; tail -32 $O/profile_fp16.log
timeout 600 python bench.py --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "bench fp32 rc=$?"
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc=$?"
timeout 600 python bench.py --dtype fp32 --eps-dtype fp16 --no-cpu-baseline > $O/bench_fp32_fp16.json 2> $O/bench_fp32_fp16.err; echo "bench fp32/fp16 rc=$?"
# the stage kernel inside the torch network loop: events for both networks, rocprofv3 rows (fp16 and fp32 state)
timeout 600 python tools/in_loop.py --kinds gemm,conv --out $O/in_loop.json > $O/in_loop.log 2>&1; echo "in_loop rc=$?"
for DT in fp16 fp32; do
  timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_loop_$DT -o kt -- python tools/in_loop.py --dtype $DT --trace-only > $O/kt_loop_$DT.log 2>&1; echo "rocprof in-loop $DT rc=$?"
  python tools/in_loop.py --summarise $O/kt_loop_$DT --md $O/in_loop_trace_$DT.md > /dev/null 2>&1
  find $O/kt_loop_$DT -name "*kernel_stats.csv" -exec cp {} $O/in_loop_kernel_stats_$DT.csv \;
  rm -rf $O/kt_loop_$DT
  sed -n 5,9p $O/in_loop_trace_$DT.md
done
for DT in fp16 fp32; do   # the same behind the conv network (MIOpen kernels in front of the stage kernel)
  timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_loopc_$DT -o kt -- python tools/in_loop.py --dtype $DT --kinds conv --trace-only > $O/kt_loopc_$DT.log 2>&1; echo "rocprof in-loop conv $DT rc=$?"
  python tools/in_loop.py --summarise $O/kt_loopc_$DT --md $O/in_loop_trace_conv_$DT.md --title "stage kernel inside a torch network loop, conv network (rocprofv3 --kernel-trace)" > /dev/null 2>&1
  rm -rf $O/kt_loopc_$DT
  sed -n 5,9p $O/in_loop_trace_conv_$DT.md
done
timeout 900 python tools/stage_bench.py --md $O/stage_table.md > $O/stage_bench.log 2>&1; echo "stage_bench rc=$?"; tail -8 $O/stage_bench.log
timeout 300 python tools/thr_routes.py > $O/thr_routes.txt 2>&1; echo "thr_routes rc=$?"
# the unmodified reference, when it travelled here as git-ignored scratch (_refscratch/, removed after the call): its CPU timing
# on this box's host cores and its timing + parity on this GPU
if [ -f _refscratch/dpm_solver_pytorch.py ]; then
  DPM_REFERENCE_DIR=_refscratch timeout 400 python tools/cpu_baseline.py --budget 40 --out $O/cpu_baseline_reference_gpubox.json --where "MI355X box host cores (gpurun)" > $O/cpu_baseline.log 2>&1; echo "cpu_baseline rc=$?"
  DPM_REFERENCE_DIR=_refscratch timeout 400 python tools/gpu_reference.py --out $O/gpu_reference.json > $O/gpu_reference.log 2>&1; echo "gpu_reference rc=$?"
fi
# the reference's own example call sites, unchanged, on the engine (needs the scratch tree)
if [ -d _refscratch/examples ]; then
  DPM_REFERENCE_DIR=_refscratch timeout 300 python tools/dropin_examples.py --device cuda:0 --out $O/dropin.json > $O/dropin.log 2>&1; echo "dropin rc=$?"; tail -4 $O/dropin.log
fi
# the other BASELINE kernels inside a real torch network loop (rocprofv3 rows), channels_last vs default layout
for CASE in nchw nhwc cfg_sd64 cfg_sd8 cfg5 cfg3; do
  TR=6; [ $CASE = cfg3 ] && TR=3
  timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_$CASE -o kt -- python tools/in_loop.py --case $CASE --trajectories $TR > $O/case_$CASE.log 2>&1; echo "rocprof $CASE rc=$?"
  grep '^{' $O/case_$CASE.log | tail -1 | cut -c1-400
  python tools/in_loop.py --summarise $O/kt_$CASE --md $O/in_loop_$CASE.md --title "BASELINE kernel inside a torch network loop: case $CASE (rocprofv3 --kernel-trace --stats -- python tools/in_loop.py --case $CASE)" > /dev/null 2>&1
  find $O/kt_$CASE -name "*kernel_stats.csv" -exec cp {} $O/in_loop_${CASE}_kernel_stats.csv \;
  rm -rf $O/kt_$CASE
  sed -n 5,9p $O/in_loop_$CASE.md | cut -c1-200
done
# cfg5's own size under the profiler: kernel rows + memory-side traffic (separate --pmc passes)
P=$O/prof_thr32; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $P/kt -o kt -- python tools/stage_bench.py --only "cfg5 2M++ thr B=32" > $P/kt.log 2>&1; echo "rocprof thr32 kt rc=$?"
find $P/kt -name "*kernel_stats.csv" -exec cp {} $P/kernel_stats.csv \;
find $P/kt -name "*kernel_trace.csv" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $P/pmc_$C -o pmc -- python tools/stage_bench.py --only "cfg5 2M++ thr B=32" > $P/pmc_$C.log 2>&1; echo "pmc $C rc=$?"
done
python tools/rocprof_summary.py $P "stage_thresh_kernel<float, float, 1, 0, false, 512, 1" $P/summary.md "$TAG: rocprofv3 ... -- python tools/stage_bench.py --only 'cfg5 2M++ thr B=32' (cfg5's own size [32,3,64,64])" > /dev/null 2>&1
tail -12 $P/summary.md
find $O -name "*.db" -size +20M -delete
du -sh $O
