#!/bin/bash
# round 5, call U: experiment -- the library built with --offload-compress (4.7 MB instead of 13.6): does it load, and what does the
# decompression cost at cold start?  (tools/_variants/zc, built by __graft_entry__.build_variant('zc', ['--offload-compress']))
TAG=${1:-r05u}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
ls -la dpm_solver_amd/libdpm_hip.so tools/_variants/zc/libdpm_hip.so
for rep in 1 2; do
  timeout 600 python tools/cold_start.py --repeat 3 --out $O/cold_plain_$rep.json > $O/cold_plain_$rep.log 2>&1; echo "plain $rep rc=$?"; grep -o '"scenario": "[a-z]*"\|"cold_start_ms": [0-9.]*\|"import_dpm_solver_amd_ms": [0-9.]*\|"first_sample_ms": [0-9.]*' $O/cold_plain_$rep.log | paste - - - - | head -3
  DPM_SOLVER_AMD_LIB=$GRAFT_REPO_ROOT/tools/_variants/zc/libdpm_hip.so timeout 600 python tools/cold_start.py --repeat 3 --out $O/cold_zc_$rep.json > $O/cold_zc_$rep.log 2>&1; echo "compressed $rep rc=$?"; grep -o '"scenario": "[a-z]*"\|"cold_start_ms": [0-9.]*\|"import_dpm_solver_amd_ms": [0-9.]*\|"first_sample_ms": [0-9.]*' $O/cold_zc_$rep.log | paste - - - - | head -3
done
( time DPM_SOLVER_AMD_LIB=$GRAFT_REPO_ROOT/tools/_variants/zc/libdpm_hip.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_cover.py -m gpu -q -x ) > $O/pytest_zc.log 2>&1; echo "pytest on the compressed library rc=$?"; tail -3 $O/pytest_zc.log
