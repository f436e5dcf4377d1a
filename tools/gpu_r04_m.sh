#!/bin/bash
# round 4, call m: the regression test of the reworked recovery + memory-system ceilings of 3R+2W against 4R+1W
TAG=${1:-r04m}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "staggered or timeout" > $O/pytest_fault.log 2>&1; echo "fault tests rc=$?"; tail -2 $O/pytest_fault.log
timeout 900 python tools/calib.py > $O/calib.jsonl 2>$O/calib.err; echo "calib rc=$?"
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r04m/calib.jsonl") if l.startswith("{")]
best={}
for r in rows:
    for mode in ("warm","cold"):
        k=(r["size"],r["kind"],mode)
        if k not in best or r[mode+"_GBs"]>best[k][mode+"_GBs"]: best[k]=r
for k in sorted(best): r=best[k]; print(k, r[k[2]+"_GBs"], "GB/s", r[k[2]+"_us"], "us  block",r["block"],"bpc",r["bpc"],"nt",r["nt"])
PY
