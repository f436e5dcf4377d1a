#!/bin/bash
# round 5, call L: the default bench line of the trimmed library (with cold_start_ms), and the driver's torchrun form
TAG=${1:-r05l}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc=$?"; tail -3 $O/bench_default.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun_1rank.err; echo "torchrun 1 rank rc=$?"
python - <<PY
import json
for f in ("bench_default","bench_torchrun_1rank"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["value"], d["roofline"]["frac"], d["launcher"], d["ms_per_step"], d["roofline"].get("in_network_loop",{}).get("frac_of_floor")); print("  cold", d.get("cold_start_ms"))
PY
