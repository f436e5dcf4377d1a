#!/bin/bash
# round 3, call g: the predicted select bound (restricted to K <= 128): thresholding tests, long random sweeps, timeline, rows
TAG=${1:-r03g}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_extensions.py -m gpu -q -x -k "thresh or predict or cluster or thr" > $O/pytest_thr.log 2>&1; echo "pytest thr rc=$?"; tail -4 $O/pytest_thr.log
( time DPM_THR_SWEEP=4000 DPM_THR_SWEEP_STEPS=20 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_sweep" ) > $O/pytest_sweep_long.log 2>&1; echo "sweep (4000 configs, up to 19 steps) rc=$?"; tail -6 $O/pytest_sweep_long.log
timeout 600 python tools/thr_routes.py > $O/thr_routes.txt 2>&1; echo "routes rc=$?"; grep -A6 "shape (32" $O/thr_routes.txt | head -30
timeout 600 python tools/thr_timeline.py --build > $O/thr_timeline_build.log 2>&1
timeout 300 python tools/thr_timeline.py --run --batch 32 --chw 3 64 64 > $O/thr_timeline_b32.txt 2>&1; echo "timeline rc=$?"; cat $O/thr_timeline_b32.txt | head -30
timeout 600 python tools/stage_bench.py --only "thr" --md $O/stage_thr.md > $O/stage_thr.log 2>&1; echo "stage_bench thr rc=$?"; grep -i "thr" $O/stage_thr.md | head -14
