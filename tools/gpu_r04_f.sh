#!/bin/bash
# round 4, call f: A/B on one box -- 128-byte aligned slots (288 words) vs round 3's 264-word slots, thresholding rows
TAG=${1:-r04f}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for REP in 1 2; do
  timeout 600 python tools/stage_bench.py --only "thr" --md $O/stage_thr_aligned_$REP.md > $O/a_$REP.log 2>&1; echo "aligned $REP rc=$?"; grep -E "TWO - thr \+m" $O/stage_thr_aligned_$REP.md
  DPM_SOLVER_AMD_LIB=tools/_variants/slot264/libdpm_hip.so timeout 600 python tools/stage_bench.py --only "thr" --md $O/stage_thr_slot264_$REP.md > $O/b_$REP.log 2>&1; echo "slot264 $REP rc=$?"; grep -E "TWO - thr \+m" $O/stage_thr_slot264_$REP.md
done
