#!/bin/bash
# round 5, call T: the runs that need the reference's files on the box (they travel as git-ignored scratch, tools/ref_scratch.sh,
# removed right after the call): the three drop-in GPU tests, the reference's own example call sites through shims/
# (tools/dropin_examples.py), the unmodified reference on the same MI355X next to the engine (tools/gpu_reference.py), the
# reference on this box's host cores (tools/cpu_baseline.py), and the bench line with a LIVE reference cpu_baseline
TAG=${1:-r05t}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
export DPM_REFERENCE_DIR=$GRAFT_REPO_ROOT/_refscratch
ls $DPM_REFERENCE_DIR | head -3
( time timeout 600 python -m pytest tests/test_gpu_extensions.py -m gpu -q -k "dropin_examples_on_the_gpu" ) > $O/pytest_dropin.log 2>&1; echo "drop-in tests rc=$?"; tail -4 $O/pytest_dropin.log
timeout 600 python tools/dropin_examples.py --device cuda:0 --out $O/dropin.json > $O/dropin.log 2>&1; echo "dropin rc=$?"; tail -6 $O/dropin.log
timeout 600 python tools/gpu_reference.py --out $O/gpu_reference.json > $O/gpu_reference.log 2>&1; echo "gpu_reference rc=$?"; tail -8 $O/gpu_reference.log
timeout 600 python tools/cpu_baseline.py --out $O/cpu_baseline_reference_gpubox.json --where "MI355X box host cores (gpurun)" --budget 30 > $O/cpu_baseline.log 2>&1; echo "cpu_baseline rc=$?"; tail -4 $O/cpu_baseline.log
( time timeout 900 python bench.py > $O/bench_live_reference.json 2> $O/bench_live_reference.err ); echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_live_reference.json")); c=d["cpu_baseline"]; print(d["value"], d["roofline"]["frac"]); print({k:c[k] for k in ("value","unit","cores","kind","measured_in_this_run") if k in c}); print(c.get("sample","")[:200])
PY
