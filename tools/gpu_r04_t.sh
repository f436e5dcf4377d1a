#!/bin/bash
# round 4, call t: the sample mark polled by one wavefront per workgroup -- fault tests and sweeps, cfg5's size under the
# profiler (rows + memory-side traffic), cfg5 inside the loop
TAG=${1:-r04t}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "timeout or staggered or thresholding or cluster" > $O/pytest_thr.log 2>&1; echo "thresholding tests rc=$?"; tail -1 $O/pytest_thr.log
for CASE in "1 1" "3 1" "2 0"; do
  set -- $CASE
  DPM_THR_SWEEP_FAULT=$1 DPM_THR_SWEEP_ONE_HOP=$2 DPM_THR_SWEEP=3000 DPM_THR_SWEEP_STEPS=12 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_sweep" > $O/sweep_fault_$1_onehop_$2.log 2>&1
  echo "sweep of 3000, fault mode $1, one_hop $2: rc=$?  $(grep -E 'passed|failed' $O/sweep_fault_$1_onehop_$2.log | tail -1)"
done
P=$O/prof_thr32; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $P/kt -o kt -- python tools/stage_bench.py --only "cfg5 2M++ thr B=32" > $P/kt.log 2>&1; echo "rocprof thr32 kt rc=$?"
find $P/kt -name "*kernel_stats.csv" -exec cp {} $P/kernel_stats.csv \;
find $P/kt -name "*kernel_trace.csv" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $P/pmc_$C -o pmc -- python tools/stage_bench.py --only "cfg5 2M++ thr B=32" > $P/pmc_$C.log 2>&1; echo "pmc $C rc=$?"
done
python tools/rocprof_summary.py $P "stage_thresh_kernel<float, float, 1, 0, false, 512, 1" $P/summary.md "$TAG: rocprofv3 ... -- python tools/stage_bench.py --only 'cfg5 2M++ thr B=32' (cfg5's own size [32,3,64,64])" > /dev/null 2>&1
tail -9 $P/summary.md
timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_cfg5 -o kt -- python tools/in_loop.py --case cfg5 --trajectories 6 > $O/case_cfg5.log 2>&1
python tools/in_loop.py --summarise $O/kt_cfg5 --md $O/in_loop_cfg5.md --title "BASELINE kernel inside a torch network loop: case cfg5" > /dev/null 2>&1
find $O/kt_cfg5 -name "*kernel_stats.csv" -exec cp {} $O/in_loop_cfg5_kernel_stats.csv \;
rm -rf $O/kt_cfg5; sed -n 5,9p $O/in_loop_cfg5.md | cut -c1-200
find $O -name "*.db" -size +20M -delete
