#!/bin/bash
# round 5, call H: the double-precision kernels after their rework (elements in flight, LDS candidate list)
TAG=${1:-r05h}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "double" ) > $O/pytest_double.log 2>&1; echo "pytest double rc=$?"; tail -5 $O/pytest_double.log
timeout 600 python tools/f64_bench.py --out $O/f64.json > $O/f64.log 2>&1; echo "f64 rc=$?"; tail -7 $O/f64.log
