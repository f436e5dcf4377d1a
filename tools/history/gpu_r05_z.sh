#!/bin/bash
# round 5, last call: the tree as committed -- full GPU suite, smoke, the stage kernel inside the conv-network loop by rocprofv3
# rows (fp16 / fp32) on the last library, the default bench line
TAG=${1:-r05z}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
export DPM_SOLVER_AMD_LIB_SAVED=$DPM_SOLVER_AMD_LIB
for DT in fp16 fp32; do
  DPM_SOLVER_AMD_LIB=tools/_variants/lab/libdpm_lab.so timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_loopc_$DT -o kt -- python tools/in_loop.py --dtype $DT --kinds conv --trace-only > $O/kt_loopc_$DT.log 2>&1; echo "rocprof in-loop conv $DT rc=$?"
  python tools/in_loop.py --summarise $O/kt_loopc_$DT --md $O/in_loop_trace_conv_$DT.md --title "stage kernel inside a torch network loop, conv network (rocprofv3 --kernel-trace), the round's last library" > /dev/null 2>&1
  rm -rf $O/kt_loopc_$DT
  sed -n 5,9p $O/in_loop_trace_conv_$DT.md
done
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); inl=d["roofline"].get("in_network_loop",{}); print(d["value"], d["roofline"]["frac"], d["ms_per_step"], inl.get("stage_kernel_us"), inl.get("frac_of_floor")); print(d.get("cold_start_ms",{}).get("plain_8x4x64x64"), d.get("cold_start_ms",{}).get("network_warm"))
PY
