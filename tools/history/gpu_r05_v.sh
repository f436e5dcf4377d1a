#!/bin/bash
# round 5, call V: cold start with the network's own first-call cost separated (tools/cold_start.py), and the bench line carrying it
TAG=${1:-r05v}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 600 python tools/cold_start.py --repeat 3 --out $O/cold_start.json ) > $O/cold_start.log 2>&1; echo "cold start rc=$?"
python - <<PY
import json
for r in json.load(open("$O/cold_start.json"))["rows"]:
    print(r["scenario"], {k: r[k] for k in ("import_dpm_solver_amd_ms","first_sample_ms","cold_start_ms","network_first_call_ms","first_sample_network_warm_ms","cold_start_network_warm_ms","second_sample_ms")})
PY
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); print(d["value"], d["roofline"]["frac"]); print(d["cold_start_ms"])
PY
