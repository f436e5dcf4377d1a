#!/bin/bash
# round 5, call R: rocprofv3 kernel stats + memory-side counters of the headline command on the round's last library
TAG=${1:-r05r}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 bash tools/profile_round.sh $TAG fp16 > $O/profile_fp16.log 2>&1; echo "profile rc=$?"; tail -32 $O/profile_fp16.log
