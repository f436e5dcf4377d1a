#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_extensions.py tests/test_gpu_multi.py -m gpu -x -q -k "thresh or cluster or cfg5 or captured or unfused" > $O/pytest_thr.log 2>&1; echo "pytest-thr rc=$?"; tail -25 $O/pytest_thr.log
timeout 600 python tools/stage_bench.py --only thr > $O/stage_thr.log 2>&1; echo "stage rc=$?"; grep "thr" $O/stage_thr.log | tail -20
