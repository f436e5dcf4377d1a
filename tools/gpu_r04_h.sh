#!/bin/bash
# round 4, call h: bisect of the clustered thresholding rows on ONE box: round 3's tree, the tree after the fault-recovery
# change (201176e), this tree, and this tree with the old clean-up / without the split rank counting
TAG=${1:-r04h}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
row() { grep -E "TWO - thr \+m" $1 | grep -v "6ch" | awk -F'|' '{printf "%s b2b %s evicted %s; ", $2, $6, $9}'; echo; }
for REP in 1 2; do
  ( cd _r03tree && timeout 600 python tools/stage_bench.py --only "thr" --md $O/r03_$REP.md > /dev/null 2>&1 ); echo -n "r03      $REP: "; row $O/r03_$REP.md
  ( cd _v1tree && timeout 600 python tools/stage_bench.py --only "thr" --md $O/v1_$REP.md > /dev/null 2>&1 ); echo -n "v1(201176e) $REP: "; row $O/v1_$REP.md
  timeout 600 python tools/stage_bench.py --only "thr" --md $O/cur_$REP.md > /dev/null 2>&1; echo -n "current  $REP: "; row $O/cur_$REP.md
  DPM_SOLVER_AMD_LIB=tools/_variants/oldclean/libdpm_hip.so timeout 600 python tools/stage_bench.py --only "thr" --md $O/oldclean_$REP.md > /dev/null 2>&1; echo -n "oldclean $REP: "; row $O/oldclean_$REP.md

done
