#!/bin/bash
TAG=${1:-r04q2}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
row() {  # case block_threads tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_$3 -o kt -- python tools/in_loop.py --case $1 --trajectories 6 --block-threads $2 > $O/kt_$3.log 2>&1
  python tools/in_loop.py --summarise $O/kt_$3 --md $O/in_loop_$3.md --title "case $1, block_threads knob $2" > /dev/null 2>&1
  rm -rf $O/kt_$3
  echo "$1 bt=$2: $(grep 'stage_kernel<' $O/in_loop_$3.md | sort -t'|' -k3 -n -r | head -1 | sed 's/.*` |//' | tr '\n' ';')"
}
for REP in 1 2 3 4 5 6; do
  for BT in 256 512 0; do row autocast256 $BT autocast256_${BT}_$REP; done
done
