#!/bin/bash
# round 5, call K: the trimmed library (478 stage kernels) -- full GPU suite twice, smoke, the default bench line (now with
# cold_start_ms), stage_bench of the product paths (lab build of the same sources: what every scenario costs now), cold start
TAG=${1:-r05k}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for i in 1 2; do
  ( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_$i.log 2>&1; echo "pytest run $i rc=$?"; grep -E "passed|failed" $O/pytest_$i.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_$i.log | head
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); print(d["value"], d["roofline"]["frac"], d["ms_per_step"], d["roofline"].get("in_network_loop",{}).get("frac_of_floor")); print("cold", d.get("cold_start_ms"))
PY
( time timeout 600 python tools/cold_start.py --repeat 3 --out $O/cold_start.json ) > $O/cold_start.log 2>&1; echo "cold start rc=$?"; grep cold_start_ms $O/cold_start.log | cut -c1-200
DPM_SOLVER_AMD_LIB=tools/_variants/lab/libdpm_lab.so timeout 900 python tools/stage_bench.py --md $O/stage_bench.md > $O/stage_bench.log 2>&1; echo "stage_bench rc=$?"
grep -E "^\| (cfg|SD|3M|uncond|guided|2M|inpaint)" $O/stage_bench.md | cut -c1-150 | head -80
