#!/bin/bash
# round 4, call g: same-box A/B of the thresholding rows -- round 3's tree (_r03tree/, scratch) vs this tree
TAG=${1:-r04g}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
for REP in 1 2; do
  ( cd _r03tree && timeout 600 python tools/stage_bench.py --only "thr" --md $O/stage_thr_r03_$REP.md > $O/r03_$REP.log 2>&1; echo "r03 $REP rc=$?"; grep -E "thr \+m|TWO - thr \|" $O/stage_thr_r03_$REP.md )
  timeout 600 python tools/stage_bench.py --only "thr" --md $O/stage_thr_r04_$REP.md > $O/r04_$REP.log 2>&1; echo "r04 $REP rc=$?"; grep -E "thr \+m|TWO - thr \|" $O/stage_thr_r04_$REP.md
done
( cd _r03tree && timeout 600 python tools/thr_routes.py > $O/thr_routes_r03.txt 2>&1 ); grep -A4 "^shape (32" $O/thr_routes_r03.txt | head -12
timeout 600 python tools/thr_routes.py > $O/thr_routes_r04.txt 2>&1; grep "^shape" $O/thr_routes_r04.txt
