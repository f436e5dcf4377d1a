#!/bin/bash
# round 5, call W: which kernels the FIRST sample() of a process launches besides the library's own (tools/first_call_kernels.py)
TAG=${1:-r05w}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for s in plain thresholding cfg; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$s -o kt -- python tools/first_call_kernels.py $s > $O/$s.log 2>&1; echo "== $s rc=$?"
  python - <<PY
import csv, glob
for f in glob.glob("$O/$s/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Calls"], r["Name"][:120])
PY
done
find $O -name "*kernel_trace.csv" -delete
