#!/bin/bash
# round 5, call J: what the eps-form compile-time prologues (SPEC_NOISE_EPS) are worth -- stage_bench on the eps-form scenarios,
# specialised vs run-time prologue, alternating twice on one box
TAG=${1:-r05j}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
export DPM_SOLVER_AMD_LIB=tools/_variants/lab/libdpm_lab.so
for rep in 1 2; do
  for sc in "eps form" "cfg3 3S dpmsolver"; do
    tag=$(echo "$sc" | tr ' ' '_')
    timeout 600 python tools/stage_bench.py --only "$sc" --md $O/spec_${tag}_$rep.md > $O/spec_${tag}_$rep.log 2>&1; echo "spec $sc $rep rc=$?"
    timeout 600 python tools/stage_bench.py --only "$sc" --force-generic --md $O/generic_${tag}_$rep.md > $O/generic_${tag}_$rep.log 2>&1; echo "generic $sc $rep rc=$?"
  done
done
tail -n 12 $O/spec_eps_form_1.md; tail -n 12 $O/generic_eps_form_1.md
