#!/bin/bash
# round 4, call o: threads per workgroup of the streaming stage kernel (256 / 512 / 1024; every 256-lane group takes tiles
# of its own): bit identity, then the 2M stage INSIDE the torch network loop by rocprofv3 rows, two alternating repetitions
TAG=${1:-r04o}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "workgroup_size" > $O/pytest_bt.log 2>&1; echo "bit identity rc=$?"; tail -3 $O/pytest_bt.log
row() {  # dtype kind block_threads tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_$4 -o kt -- python tools/in_loop.py --dtype $1 --kinds $2 --trace-only --block-threads $3 > $O/kt_$4.log 2>&1
  python tools/in_loop.py --summarise $O/kt_$4 --md $O/in_loop_$4.md --title "2M stage kernel in the $2 loop, $1, $3 threads per workgroup" > /dev/null 2>&1
  rm -rf $O/kt_$4
  echo "$1 $2 bt=$3: $(grep 'stage_kernel<' $O/in_loop_$4.md | grep -v Li0ELi0 | head -3 | sed 's/.*` |//' | tr '\n' ';')"
}
for REP in 1 2; do
  for BT in 256 512 1024; do row fp16 conv $BT fp16_conv_${BT}_$REP; done
done
for BT in 256 512 1024; do row fp32 conv $BT fp32_conv_${BT}_1; done
for BT in 256 1024; do row fp16 gemm $BT fp16_gemm_${BT}_1; done
