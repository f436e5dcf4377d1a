#!/bin/bash
# round 4, call l: the reworked cluster-fault recovery -- larger forced-fault sweeps, same-box A/B of the thresholding
# stages against the previous library (tools/_variants/prev, scratch), the whole GPU suite
TAG=${1:-r04l}
N=${2:-10000}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for CASE in "0 1" "1 1" "3 1" "1 0" "2 0"; do
  set -- $CASE
  ( time DPM_THR_SWEEP_FAULT=$1 DPM_THR_SWEEP_ONE_HOP=$2 DPM_THR_SWEEP=$N DPM_THR_SWEEP_STEPS=12 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_sweep" ) > $O/sweep_fault_$1_onehop_$2.log 2>&1
  echo "sweep of $N with forced faults mode $1, one_hop $2: rc=$?  $(grep -E "passed|failed" $O/sweep_fault_$1_onehop_$2.log | tail -1)  $(grep real $O/sweep_fault_$1_onehop_$2.log)"
done
if [ -f tools/_variants/prev/libdpm_hip.so ]; then
  for R in 1 2; do
    DPM_SOLVER_AMD_LIB=$PWD/tools/_variants/prev/libdpm_hip.so timeout 300 python tools/thr_catchall_ab.py --label prev_$R > $O/ab_prev_$R.log 2>&1; tail -12 $O/ab_prev_$R.log
    timeout 300 python tools/thr_catchall_ab.py --label new_$R > $O/ab_new_$R.log 2>&1; tail -12 $O/ab_new_$R.log
  done
fi
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -4 $O/pytest_gpu.log
