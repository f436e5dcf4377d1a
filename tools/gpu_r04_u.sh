#!/bin/bash
# round 4, call u: is the two-workgroups-per-CU condition of the 512-thread rule right?  SD-size CFG launches (512 tiles: the
# default keeps 256 threads) with 512 forced, alternating
TAG=${1:-r04u}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for REP in 1 2 3; do
  for BT in 256 512; do
    T=cfg_sd64_${BT}_$REP
    timeout 300 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_$T -o kt -- python tools/in_loop.py --case cfg_sd64 --trajectories 6 --block-threads $BT > $O/kt_$T.log 2>&1
    python tools/in_loop.py --summarise $O/kt_$T --md $O/in_loop_$T.md --title "case cfg_sd64, block_threads knob $BT" > /dev/null 2>&1
    rm -rf $O/kt_$T
    echo "cfg_sd64 bt=$BT rep $REP: $(grep 'stage_kernel<' $O/in_loop_$T.md | grep '| 144 |' | sed 's/.*` | 144 | //' | cut -d'|' -f1,2)"
  done
done
