#!/bin/bash
# round 4, call i: specialised thresholding kernels also for networks that are not plain noise predictors (generic prologue
# behind a wave-uniform branch) vs the catch-all kernel, same box; quickstart test
TAG=${1:-r04i}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_extensions.py -m gpu -q -x -k "quickstart" > $O/pytest_q.log 2>&1; echo "quickstart rc=$?"; tail -2 $O/pytest_q.log
for REP in 1 2; do
  timeout 300 python tools/thr_catchall_ab.py --label default_$REP > $O/ab_default_$REP.jsonl 2> $O/ab_default_$REP.err; echo "default rc=$?"
  DPM_SOLVER_AMD_LIB=tools/_variants/prev/libdpm_hip.so timeout 300 python tools/thr_catchall_ab.py --label prev_$REP > $O/ab_prev_$REP.jsonl 2> $O/ab_prev_$REP.err; echo "prev rc=$?"
done
paste -d'\n' $O/ab_default_1.jsonl $O/ab_prev_1.jsonl $O/ab_default_2.jsonl $O/ab_prev_2.jsonl | python -c "
import sys, json, collections
rows = collections.OrderedDict()
for l in sys.stdin:
    r = json.loads(l); rows.setdefault((tuple(r['shape']), r['network']), {}).setdefault(r['label'].split('_')[0], []).append(r['us_per_stage_median'])
for k, v in rows.items(): print(k, {a: b for a, b in v.items()})
"
DPM_THR_SWEEP=4000 DPM_THR_SWEEP_STEPS=20 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_extensions.py -m gpu -q -x -k "thresh or thr or cfg5 or guided or random_configurations or random_sweep or cluster" > $O/pytest_final.log 2>&1; echo "pytest final rc=$?"; tail -2 $O/pytest_final.log
# kernel-only rows: this library vs the previous commit's, same box
for REP in 1 2; do
  timeout 600 python tools/stage_bench.py --only "thr" --md $O/stage_new_$REP.md > /dev/null 2>&1; echo -n "new  $REP: "; grep -E "TWO - thr \+m" $O/stage_new_$REP.md | awk -F'|' '{printf "%s %s /%s; ", $2, $6, $9}'; echo
  DPM_SOLVER_AMD_LIB=tools/_variants/prev/libdpm_hip.so timeout 600 python tools/stage_bench.py --only "thr" --md $O/stage_prev_$REP.md > /dev/null 2>&1; echo -n "prev $REP: "; grep -E "TWO - thr \+m" $O/stage_prev_$REP.md | awk -F'|' '{printf "%s %s /%s; ", $2, $6, $9}'; echo
done
