#!/bin/bash
# round 5, call N: the new HOT 3 coverage test, then the name set of the whole GPU suite once more (tools/kernel_names.py)
TAG=${1:-r05n}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "other_parameterisations_specialised" ) > $O/hot3.log 2>&1; echo "hot3 rc=$?"; tail -5 $O/hot3.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_e2e -o kt -- python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_e2e_vs_reference_goldens_and_oracle" > $O/e2e.log 2>&1; echo "e2e rc=$?"
find $O/kt_e2e -name "*kernel_trace.csv" -delete
python tools/kernel_names.py --summarise $O/kt_e2e --md $O/kernels_launched_e2e.md --title "stage kernels launched by the 51 end-to-end golden cases (tests/test_gpu_parity.py::test_e2e_vs_reference_goldens_and_oracle, rocprofv3 --kernel-trace --stats)"
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_suite -o kt -- python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_extensions.py::test_lab_suite_on_the_lab_build > $O/suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $O/suite.log | tail -1
find $O/kt_suite -name "*kernel_trace.csv" -delete
python tools/kernel_names.py --summarise $O/kt_suite --md $O/kernels_launched_suite.md --title "stage kernels launched by the GPU suite on the product library (pytest tests -m gpu under rocprofv3 --kernel-trace --stats)"
find $O -name "*.db" -size +10M -delete
