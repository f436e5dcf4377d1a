#!/bin/bash
# round 5, call A: the refactored product / lab libraries on the GPU -- suite (product + lab subprocess), the floor sweep of
# the lone launch (item 1), specialised vs run-time-prologue kernels (kernel-count budget, item 4), cold start, the bench line
TAG=${1:-r05a}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|skipped|real" $O/pytest.log | tail -3
timeout 300 python tools/floor.py --check > $O/floor_check.log 2>&1; echo "floor check rc=$?"; tail -2 $O/floor_check.log
( time timeout 600 python tools/floor.py --sweep --out $O/floor.json ) > $O/floor_sweep.log 2>&1; echo "floor sweep rc=$?"; tail -18 $O/floor_sweep.log
for REP in 1 2; do
  timeout 600 python tools/stage_bench.py --md $O/stage_spec_$REP.md > $O/stage_spec_$REP.log 2>&1; echo "stage_bench specialised $REP rc=$?"
  timeout 600 python tools/stage_bench.py --force-generic --md $O/stage_generic_$REP.md > $O/stage_generic_$REP.log 2>&1; echo "stage_bench generic $REP rc=$?"
done
( time timeout 600 python tools/cold_start.py --repeat 3 --out $O/cold_start.json ) > $O/cold_start.log 2>&1; echo "cold start rc=$?"; cat $O/cold_start.log | tail -6
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc=$?"; tail -c 1500 $O/bench_default.json; tail -5 $O/bench_default.err
