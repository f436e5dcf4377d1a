#!/bin/bash
# round 5, call G: the tree as committed -- full GPU suite, smoke, the double-precision path's own numbers
TAG=${1:-r05g}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|skipped|real" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python tools/f64_bench.py --out $O/f64.json > $O/f64.log 2>&1; echo "f64 rc=$?"; cat $O/f64.log | tail -12
