#!/bin/bash
# round 4, call v: forced faults with clusters that walk several samples ([300,1,128,128]: 600 workgroups in clusters of two)
TAG=${1:-r04v}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "timeout_is_recovered or staggered" > $O/pytest_fault.log 2>&1; echo "fault tests rc=$?"; tail -3 $O/pytest_fault.log
for CASE in "0 1" "1 1" "3 1" "2 0" "1 0"; do
  set -- $CASE
  DPM_THR_SWEEP_FAULT=$1 DPM_THR_SWEEP_ONE_HOP=$2 DPM_THR_SWEEP=1200 DPM_THR_SWEEP_STEPS=10 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_sweep" > $O/sweep_fault_$1_onehop_$2.log 2>&1
  echo "sweep of 1200 (30 of them [300,1,128,128]), fault mode $1, one_hop $2: rc=$?  $(grep -E 'passed|failed' $O/sweep_fault_$1_onehop_$2.log | tail -1)"
done
