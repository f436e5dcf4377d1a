#!/bin/bash
# round 3, call d: the fat-wavefront variants inside the torch network loop for an fp32 state (rocprofv3 rows + events)
TAG=${1:-r03d}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
export DPM_SOLVER_AMD_LIB=$PWD/tools/_variants/tune/libdpm_hip.so
for U in 1 2 4 8; do
  timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt32_u$U -o kt -- python tools/in_loop.py --dtype fp32 --trace-only --unroll $U --nt 5 > $O/kt32_u$U.log 2>&1; echo "rocprof fp32 U=$U rc=$?"
  python tools/in_loop.py --summarise $O/kt32_u$U --md $O/in_loop_trace32_u$U.md > /dev/null 2>&1
  find $O/kt32_u$U -name "*.db" -delete; find $O/kt32_u$U -name "*kernel_trace.csv" -delete
  sed -n 5,9p $O/in_loop_trace32_u$U.md
  tail -1 $O/kt32_u$U.log | grep -o '"stage_kernel_us": [0-9.]*\|"stage_added_wall_us": [0-9.]*'
done
for U in 1 8; do
  timeout 300 python tools/in_loop.py --dtype fp16 --trace-only --unroll $U --nt 1 2>/dev/null | grep -o '"stage_kernel_us": [0-9.]*\|"stage_added_wall_us": [0-9.]*\|"trajectory_ms": [0-9.]*'; echo "fp16 events U=$U"
done
du -sh $O
