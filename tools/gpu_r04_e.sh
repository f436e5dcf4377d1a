#!/bin/bash
# round 4, call e: rank counting spread over all eight wavefronts -- correctness + routes + stage rows
TAG=${1:-r04e}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_extensions.py -m gpu -q -x -k "thresh or predict or cluster or thr or cfg5" > $O/pytest_thr.log 2>&1; echo "pytest thr rc=$?"; tail -3 $O/pytest_thr.log
( time DPM_THR_SWEEP=4000 DPM_THR_SWEEP_STEPS=20 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_sweep" ) > $O/pytest_sweep_long.log 2>&1; echo "sweep rc=$?"; tail -5 $O/pytest_sweep_long.log | head -2
timeout 600 python tools/thr_routes.py > $O/thr_routes.txt 2>&1; echo "routes rc=$?"; grep "^shape" $O/thr_routes.txt
timeout 600 python tools/stage_bench.py --only "thr" --md $O/stage_thr.md > $O/stage_thr.log 2>&1; echo "stage_bench thr rc=$?"; grep -i "thr" $O/stage_thr.md | sed -n 2,14p
