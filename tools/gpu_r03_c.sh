#!/bin/bash
# round 3, third GPU call: fewer, fatter wavefronts for the lone cold launch (U = 4 / 8 tiles per workgroup, all loads up
# front): tune.py single (events, cache-resident and from HBM) and the winner inside the torch network loop (rocprofv3 rows)
TAG=${1:-r03c}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
export DPM_SOLVER_AMD_LIB=$PWD/tools/_variants/tune/libdpm_hip.so
timeout 600 python tools/tune.py single --dtypes fp16,fp32 --reps 5 > $O/tune_single.txt 2> $O/tune_single.err; echo "tune rc=$?"; cat $O/tune_single.txt
for U in 1 4 8; do
  timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_u$U -o kt -- python tools/in_loop.py --trace-only --unroll $U --nt 1 > $O/kt_u$U.log 2>&1; echo "rocprof U=$U rc=$?"
  python tools/in_loop.py --summarise $O/kt_u$U --md $O/in_loop_trace_u$U.md > /dev/null 2>&1
  find $O/kt_u$U -name "*kernel_stats.csv" -exec cp {} $O/in_loop_kernel_stats_u$U.csv \;
  find $O/kt_u$U -name "*.db" -delete; find $O/kt_u$U -name "*kernel_trace.csv" -delete
  sed -n 5,9p $O/in_loop_trace_u$U.md
done
du -sh $O
