#!/bin/bash
# round 4, call a: GPU suite + the reference's example call sites unchanged on the engine (needs _refscratch/)
TAG=${1:-r04a}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
export DPM_REFERENCE_DIR=_refscratch
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 300 python tools/dropin_examples.py --device cuda:0 --out $O/dropin.json > $O/dropin.log 2>&1; echo "dropin rc=$?"; tail -5 $O/dropin.log
