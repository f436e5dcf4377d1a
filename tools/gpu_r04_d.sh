#!/bin/bash
# round 4, call d: the predicted-bound select by one wavefront (cluster_select_wave0) + precise workspace clean-up:
# correctness (thresholding tests, fault recovery, long random sweep), routes, timeline, stage rows, rocprofv3 rows + traffic
TAG=${1:-r04d}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_extensions.py -m gpu -q -x -k "thresh or predict or cluster or thr or cfg5" > $O/pytest_thr.log 2>&1; echo "pytest thr rc=$?"; tail -4 $O/pytest_thr.log
( time DPM_THR_SWEEP=4000 DPM_THR_SWEEP_STEPS=20 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_sweep" ) > $O/pytest_sweep_long.log 2>&1; echo "sweep (4000 configs, up to 19 steps) rc=$?"; tail -6 $O/pytest_sweep_long.log
timeout 600 python tools/thr_routes.py > $O/thr_routes.txt 2>&1; echo "routes rc=$?"; grep "^shape" $O/thr_routes.txt
timeout 600 python tools/thr_timeline.py --build > $O/thr_timeline_build.log 2>&1
timeout 300 python tools/thr_timeline.py --run --batch 32 --chw 3 64 64 > $O/thr_timeline_b32.txt 2>&1; echo "timeline rc=$?"; cat $O/thr_timeline_b32.txt | head -30
timeout 600 python tools/stage_bench.py --only "thr" --md $O/stage_thr.md > $O/stage_thr.log 2>&1; echo "stage_bench thr rc=$?"; grep -i "thr" $O/stage_thr.md | head -14
P=$O/prof_thr32; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $P/kt -o kt -- python tools/stage_bench.py --only "cfg5 2M++ thr B=32" > $P/kt.log 2>&1; echo "rocprof thr32 kt rc=$?"
find $P/kt -name "*kernel_stats.csv" -exec cp {} $P/kernel_stats.csv \;
find $P/kt -name "*kernel_trace.csv" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $P/pmc_$C -o pmc -- python tools/stage_bench.py --only "cfg5 2M++ thr B=32" > $P/pmc_$C.log 2>&1; echo "pmc $C rc=$?"
done
python tools/rocprof_summary.py $P "stage_thresh_kernel<float, float, 1, 0, false, 512, 1" $P/summary.md "$TAG: rocprofv3 ... -- python tools/stage_bench.py --only 'cfg5 2M++ thr B=32' (cfg5's own size [32,3,64,64])" > /dev/null 2>&1
tail -12 $P/summary.md
find $P -name "*.db" -size +20M -delete
timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_cfg5 -o kt -- python tools/in_loop.py --case cfg5 --trajectories 6 > $O/case_cfg5.log 2>&1; echo "rocprof in-loop cfg5 rc=$?"
python tools/in_loop.py --summarise $O/kt_cfg5 --md $O/in_loop_cfg5.md --title "cfg5 thresholding inside a torch network loop" > /dev/null 2>&1
find $O/kt_cfg5 -name "*kernel_stats.csv" -exec cp {} $O/in_loop_cfg5_kernel_stats.csv \;
rm -rf $O/kt_cfg5; sed -n 3,9p $O/in_loop_cfg5.md | cut -c1-200
du -sh $O
