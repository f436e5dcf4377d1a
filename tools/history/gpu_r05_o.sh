#!/bin/bash
# round 5, call O: the kernel-family coverage test (vector kernels == scalar kernel over the cross product), then the name set of
# the whole GPU suite once more
TAG=${1:-r05o}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_kernel_cover.py -m gpu -q ) > $O/cover.log 2>&1; echo "cover rc=$?"; tail -15 $O/cover.log
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_suite -o kt -- python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_extensions.py::test_lab_suite_on_the_lab_build > $O/suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $O/suite.log | tail -1
find $O/kt_suite -name "*kernel_trace.csv" -delete
python tools/kernel_names.py --summarise $O/kt_suite --md $O/kernels_launched_suite.md --title "stage kernels launched by the GPU suite on the product library (pytest tests -m gpu under rocprofv3 --kernel-trace --stats)"
find $O -name "*.db" -size +10M -delete
