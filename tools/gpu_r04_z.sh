#!/bin/bash
# round 4, final check: what the driver runs at round end -- GPU suite, smoke, the default bench line (no reference on the box)
TAG=${1:-r04z}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|skipped" $O/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04z/bench_default.json".replace("r04z", __import__("os").environ.get("TAGX","r04z"))))
cb=d["cpu_baseline"]; r=d["roofline"]; inl=r["in_network_loop"]
print("value", d["value"], "frac", r["frac"], "traffic", r["traffic"], "steps", d["steps"], "region", d["sustained"]["region_s"])
print("cpu_baseline", {k: cb.get(k) for k in ("value","kind","cores","threads","host_cores","measured_in_this_run")}, "port_live" in cb, cb.get("port_live",{}).get("value"))
print("in_loop", {k: inl.get(k) for k in ("stage_kernel_us","frac","measured_in_this_run","stage_added_wall_us")}, inl.get("live_events_incl_dispatch_offset",{}).get("stage_kernel_us"))
PY
