#!/bin/bash
# round 4, call w: the plain fp16 2M kernel at [64,4,64,64] (512 tiles, the boundary of the 512-thread rule), 256 forced
# against the default, alternating
TAG=${1:-r04w}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for REP in 1 2 3; do
  for BT in 256 0; do
    T=plain64_${BT}_$REP
    timeout 120 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_$T -o kt -- python tools/in_loop.py --case plain64 --trajectories 6 --block-threads $BT > $O/kt_$T.log 2>&1
    python tools/in_loop.py --summarise $O/kt_$T --md $O/in_loop_$T.md --title "case plain64, block_threads knob $BT" > /dev/null 2>&1
    rm -rf $O/kt_$T
    echo "plain64 bt=$BT rep $REP: $(grep 'stage_kernel<' $O/in_loop_$T.md | grep '| 152 |' | sed 's/.*` | 152 | //' | cut -d'|' -f1,2)"
  done
done
