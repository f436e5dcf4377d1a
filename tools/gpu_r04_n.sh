#!/bin/bash
# round 4, call n: the GPU suite on the refactored launchers + does the workgroup size matter for the lone launch INSIDE the
# loop?  (no-arithmetic kernel of the 2M stage's five streams in the stage kernel's place, 256 / 512 / 1024 threads)
TAG=${1:-r04n}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
for REP in 1 2; do
for SH in 256:8:1 512:4:1 1024:2:1 1024:2:5; do
  T=$(echo $SH | tr : _)_$REP
  timeout 300 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_$T -o kt -- python tools/in_loop.py --dtype fp16 --kinds conv --calib --calib-shape $SH > $O/kt_$T.log 2>&1
  python tools/in_loop.py --summarise $O/kt_$T --pattern calib_kernel --md $O/calib_in_loop_$T.md --title "no-arithmetic 3r2w kernel, shape $SH, in the conv loop" > /dev/null 2>&1
  rm -rf $O/kt_$T
  echo "shape $SH rep $REP: $(grep calib_kernel $O/calib_in_loop_$T.md | head -1 | cut -c1-40,150-)"
done
done
