#!/bin/bash
# round 5, call F: lab experiment -- thresholding clusters started out of phase (tools/thr_stagger.py)
TAG=${1:-r05f}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
LAB=tools/_variants/lab/libdpm_lab.so
( time DPM_SOLVER_AMD_LIB=$LAB timeout 1200 python tools/thr_stagger.py --out $O/thr_stagger.jsonl ) > $O/thr_stagger.log 2>&1; echo "stagger rc=$?"
python - <<PY
import json
rows=[json.loads(l) for l in open("$O/thr_stagger.jsonl")]
from collections import defaultdict
by=defaultdict(list)
for r in rows: by[tuple(r["shape"])].append(r)
for sh, rs in by.items():
    print(sh, "undisturbed", rs[0]["undisturbed_us_per_stage"], "all bit-identical", all(r["bit_identical"] for r in rs))
    for r in sorted(rs, key=lambda r: r["ratio"])[:5]: print("   groups %d offset %.1f us: %.2f us (x%.3f)" % (r["groups"], r["offset_us"], r["us_per_stage"], r["ratio"]))
PY
