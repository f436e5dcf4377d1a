#!/bin/bash
# round 5, call E: the long thresholding sweeps after the cluster clean-up change -- the product library on
# 20 000 random configurations with long trajectories (predicted route active), the lab build under the three forced
# faults and on the general route only, each asserting a zero workspace after every configuration; then the default
# bench line once more (renamed in_network_loop fields)
TAG=${1:-r05e}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
LAB=tools/_variants/lab/libdpm_lab.so
K="-k random_sweep"
( time DPM_THR_SWEEP=${SWEEP_N:-20000} DPM_THR_SWEEP_STEPS=20 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x $K ) > $O/sweep_product.log 2>&1; echo "sweep product rc=$?"; tail -4 $O/sweep_product.log
for F in 1 2 3; do
  ( time DPM_SOLVER_AMD_LIB=$LAB DPM_THR_SWEEP=${FAULT_N:-10000} DPM_THR_SWEEP_STEPS=20 DPM_THR_SWEEP_FAULT=$F timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x $K ) > $O/sweep_fault$F.log 2>&1; echo "sweep fault $F rc=$?"; tail -4 $O/sweep_fault$F.log
done
( time DPM_SOLVER_AMD_LIB=$LAB DPM_THR_SWEEP=${FAULT_N:-10000} DPM_THR_SWEEP_STEPS=20 DPM_THR_SWEEP_ONE_HOP=0 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x $K ) > $O/sweep_general.log 2>&1; echo "sweep general route rc=$?"; tail -4 $O/sweep_general.log
( time DPM_SOLVER_AMD_LIB=$LAB DPM_THR_SWEEP=${FAULT_N:-10000} DPM_THR_SWEEP_STEPS=20 DPM_THR_SWEEP_ONE_HOP=0 DPM_THR_SWEEP_FAULT=2 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x $K ) > $O/sweep_general_fault2.log 2>&1; echo "sweep general route + fault 2 rc=$?"; tail -4 $O/sweep_general_fault2.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); r=d["roofline"]; inl=r.get("in_network_loop",{})
print(d["value"], r["frac"], {k: v for k, v in inl.items() if not isinstance(v, (dict, list))})
PY
