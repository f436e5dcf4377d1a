#!/bin/bash
# round 3, call k: the no-arithmetic kernel of the same five streams in the stage kernel's place inside the torch network loop
TAG=${1:-r03k}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for DT in fp16 fp32; do
  timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_calib_$DT -o kt -- python tools/in_loop.py --dtype $DT --calib > $O/kt_calib_$DT.log 2>&1; echo "rocprof calib $DT rc=$?"
  python tools/in_loop.py --summarise $O/kt_calib_$DT --pattern calib_kernel --md $O/in_loop_calib_$DT.md --title "no-arithmetic kernel (3 read + 2 write streams, $DT-sized) in the stage kernel's place inside the torch network loop" > /dev/null 2>&1
  find $O/kt_calib_$DT -name "*kernel_stats.csv" -exec cp {} $O/in_loop_calib_kernel_stats_$DT.csv \;
  rm -rf $O/kt_calib_$DT
  sed -n 1,9p $O/in_loop_calib_$DT.md
  timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_stage_$DT -o kt -- python tools/in_loop.py --dtype $DT --trace-only > $O/kt_stage_$DT.log 2>&1; echo "rocprof stage $DT rc=$?"
  python tools/in_loop.py --summarise $O/kt_stage_$DT --md $O/in_loop_stage_$DT.md > /dev/null 2>&1
  rm -rf $O/kt_stage_$DT
  sed -n 5,9p $O/in_loop_stage_$DT.md
done
