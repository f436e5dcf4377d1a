#!/bin/bash
# round 4, call r: 512-thread workgroups by default -- bit identity, the GPU suite, the lone launches inside the loop against
# 256 forced (alternating runs, rocprofv3 rows)
TAG=${1:-r04r}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
stat() { grep 'stage_kernel<' $1 | sort -t'|' -k3,3nr | head -1 | sed 's/.*` | //' | cut -d'|' -f1-3; }
loop() {  # dtype kind block_threads tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_$4 -o kt -- python tools/in_loop.py --dtype $1 --kinds $2 --trace-only --block-threads $3 > $O/kt_$4.log 2>&1
  python tools/in_loop.py --summarise $O/kt_$4 --md $O/in_loop_$4.md --title "2M stage kernel in the $2 loop, $1, block_threads knob $3" > /dev/null 2>&1
  rm -rf $O/kt_$4; echo "$1 $2 bt=$3: rows | mean | median: $(stat $O/in_loop_$4.md)"
}
kase() {  # case block_threads tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_$3 -o kt -- python tools/in_loop.py --case $1 --trajectories 6 --block-threads $2 > $O/kt_$3.log 2>&1
  python tools/in_loop.py --summarise $O/kt_$3 --md $O/in_loop_$3.md --title "case $1, block_threads knob $2" > /dev/null 2>&1
  rm -rf $O/kt_$3; echo "$1 bt=$2: rows | mean | median: $(stat $O/in_loop_$3.md)"
}
for REP in 1 2 3; do
  for BT in 256 0; do loop fp16 conv $BT fp16_conv_${BT}_$REP; done
  for BT in 256 0; do loop fp32 conv $BT fp32_conv_${BT}_$REP; done
  for BT in 256 0; do loop fp16 gemm $BT fp16_gemm_${BT}_$REP; done
  for BT in 256 0; do kase autocast256 $BT autocast256_${BT}_$REP; done
done
for BT in 256 0; do kase cfg3 $BT cfg3_${BT}_1; done
for BT in 256 0; do kase cfg_sd64 $BT cfg_sd64_${BT}_1; done
