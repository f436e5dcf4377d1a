#!/bin/bash
# round 3, call e: launch shape of the 2M kernel tuned INSIDE the torch network loop (the emulation with interleaved
# requests mis-ranks the variants, profiles/r03_in_loop.md)
TAG=${1:-r03e}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
export DPM_SOLVER_AMD_LIB=$PWD/tools/_variants/tune/libdpm_hip.so
timeout 900 python tools/in_loop.py --sweep --kinds gemm > $O/in_loop_sweep_gemm.txt 2> $O/in_loop_sweep_gemm.err; echo "sweep gemm rc=$?"; cat $O/in_loop_sweep_gemm.txt
timeout 900 python tools/in_loop.py --sweep --kinds conv > $O/in_loop_sweep_conv.txt 2> $O/in_loop_sweep_conv.err; echo "sweep conv rc=$?"; cat $O/in_loop_sweep_conv.txt
