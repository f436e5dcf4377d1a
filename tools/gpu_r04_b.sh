#!/bin/bash
# round 4, call b: GPU suite (channels_last, fault recovery, two processes, escape hatch, drop-in) + the drop-in tool + the
# catch-all thresholding kernel at 4 vs 2 wavefronts per SIMD
TAG=${1:-r04b}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
export DPM_REFERENCE_DIR=_refscratch
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 300 python tools/dropin_examples.py --device cuda:0 --out $O/dropin.json > $O/dropin.log 2>&1; echo "dropin rc=$?"; tail -4 $O/dropin.log
timeout 300 python tools/thr_catchall_ab.py --label waves4 > $O/catchall_waves4.jsonl 2> $O/catchall_waves4.err; echo "catchall waves4 rc=$?"
DPM_SOLVER_AMD_LIB=tools/_variants/ca2/libdpm_hip.so timeout 300 python tools/thr_catchall_ab.py --label waves2 > $O/catchall_waves2.jsonl 2> $O/catchall_waves2.err; echo "catchall waves2 rc=$?"
paste -d'\n' $O/catchall_waves4.jsonl $O/catchall_waves2.jsonl
