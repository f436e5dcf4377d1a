#!/bin/bash
# round 3, call f: the predicted select bound of the clustered thresholding kernel: tests, stage-table rows
TAG=${1:-r03f}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "thresh or predict or cluster" > $O/pytest_thr.log 2>&1; echo "pytest thr rc=$?"; tail -15 $O/pytest_thr.log
timeout 600 python tools/stage_bench.py --only "thr" --md $O/stage_thr.md > $O/stage_thr.log 2>&1; echo "stage_bench thr rc=$?"; grep -i "thr" $O/stage_thr.md | head -20
python - <<'PY' > $O/stage_thr_nopredict.txt 2>&1
import subprocess, os, sys
PY
DPM_NO_PREDICT=1 timeout 600 python -c "
import sys; sys.argv=['stage_bench','--only','thr','--md','$O/stage_thr_nopredict.md']
sys.path.insert(0,'tools')
from dpm_solver_amd import _lib as L
L.lib.dpm_tuning_set(L.TUNE_THR_PREDICT, 0)
import runpy; runpy.run_path('tools/stage_bench.py', run_name='__main__')
" > $O/stage_thr_nopredict.log 2>&1; echo "stage_bench thr (no prediction) rc=$?"; grep -i "thr" $O/stage_thr_nopredict.md | head -20
