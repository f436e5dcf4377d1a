#!/bin/bash
# round 3, call i: where the kernel arguments live (HIP_FORCE_DEV_KERNARG): first-wave kernarg fetch is part of every launch's ramp
TAG=${1:-r03i}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for KA in 0 1; do
  export HIP_FORCE_DEV_KERNARG=$KA
  timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_ka$KA -o kt -- python tools/in_loop.py --trace-only > $O/kt_ka$KA.log 2>&1; echo "rocprof in-loop KERNARG=$KA rc=$?"
  python tools/in_loop.py --summarise $O/kt_ka$KA --md $O/in_loop_trace_ka$KA.md > /dev/null 2>&1
  rm -rf $O/kt_ka$KA
  sed -n 5,9p $O/in_loop_trace_ka$KA.md; grep "gap previous" $O/in_loop_trace_ka$KA.md
  tail -1 $O/kt_ka$KA.log | grep -o '"stage_kernel_us": [0-9.]*\|"stage_added_wall_us": [0-9.]*\|"trajectory_ms": [0-9.]*'
  timeout 300 python tools/stage_bench.py --only "cfg5 2M++ thr B=32" --md $O/stage_thr_ka$KA.md > /dev/null 2>&1; grep "thr B=32" $O/stage_thr_ka$KA.md
  timeout 300 python tools/stage_bench.py --only "SD 2M++ cfg" --md $O/stage_sd_ka$KA.md > /dev/null 2>&1; grep "SD 2M" $O/stage_sd_ka$KA.md
  timeout 300 python tools/stage_bench.py --only "cfg2 2M++ float16" --md $O/stage_cfg2_ka$KA.md > $O/stage_cfg2_ka$KA.log 2>&1; grep "cfg2 2M" $O/stage_cfg2_ka$KA.md; grep "python loop" $O/stage_cfg2_ka$KA.log
done
