#!/bin/bash
# round 4, call j/k: the thresholding sweep under forced cluster faults (in-kernel recovery on every configuration)
TAG=${1:-r04k}
N=${2:-2000}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for CASE in "1 1" "2 1" "3 1" "1 0" "3 0"; do
  set -- $CASE
  ( time DPM_THR_SWEEP_FAULT=$1 DPM_THR_SWEEP_ONE_HOP=$2 DPM_THR_SWEEP=$N DPM_THR_SWEEP_STEPS=12 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_sweep" ) > $O/sweep_fault_$1_onehop_$2.log 2>&1
  echo "sweep of $N with forced faults mode $1, one_hop $2: rc=$?  $(grep -E "passed|failed" $O/sweep_fault_$1_onehop_$2.log | tail -1)  $(grep real $O/sweep_fault_$1_onehop_$2.log)"
done
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fault or timeout or two_process or cluster" ) > $O/pytest_fault.log 2>&1; echo "fault tests rc=$?"; tail -3 $O/pytest_fault.log
