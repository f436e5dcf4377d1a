#!/bin/bash
# round 5, call I: soak -- the GPU suite five times over on one box (flaky tests show here, not in the driver's run), smoke,
# the default bench line and the driver's torchrun form of it
TAG=${1:-r05i}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for i in 1 2 3 4 5; do
  ( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_$i.log 2>&1; echo "pytest run $i rc=$?"; grep -E "passed|failed" $O/pytest_$i.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_$i.log | head
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc=$?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun_1rank.err; echo "torchrun 1 rank rc=$?"
python - <<PY
import json
for f in ("bench_default","bench_torchrun_1rank"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["value"], d["roofline"]["frac"], d["launcher"], d["ms_per_step"], d["roofline"].get("in_network_loop",{}).get("frac_of_floor"))
PY
