#!/bin/bash
# round 4, call p: threads per workgroup, more repetitions in the loop (256 forced against the default rule) + the stage table
# with 256 forced against the default rule on the same box
TAG=${1:-r04p}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
row() {  # dtype kind block_threads tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_$4 -o kt -- python tools/in_loop.py --dtype $1 --kinds $2 --trace-only --block-threads $3 > $O/kt_$4.log 2>&1
  python tools/in_loop.py --summarise $O/kt_$4 --md $O/in_loop_$4.md --title "2M stage kernel in the $2 loop, $1, block_threads knob $3" > /dev/null 2>&1
  rm -rf $O/kt_$4
  echo "$1 $2 bt=$3: $(grep 'stage_kernel<' $O/in_loop_$4.md | grep -v Li0ELi0 | head -2 | sed 's/.*` |//' | tr '\n' ';')"
}
for REP in 1 2 3 4; do
  for BT in 256 0; do row fp16 conv $BT fp16_conv_${BT}_$REP; done
done
for REP in 1 2; do
  for BT in 256 0; do row fp32 conv $BT fp32_conv_${BT}_$REP; done
  for BT in 256 0; do row fp16 gemm $BT fp16_gemm_${BT}_$REP; done
done
( time timeout 900 python tools/stage_bench.py --block-threads 256 --md $O/stage_table_bt256.md ) > $O/stage_bt256.log 2>&1; echo "stage table 256 rc=$?"
( time timeout 900 python tools/stage_bench.py --md $O/stage_table_default.md ) > $O/stage_default.log 2>&1; echo "stage table default rc=$?"
grep real $O/stage_default.log
