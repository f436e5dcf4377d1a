#!/bin/bash
# _refscratch/: the slice of the reference checkout the GPU-side reference runs need (tools/gpu_reference.py,
# tools/cpu_baseline.py, tools/dropin_examples.py), as git-ignored scratch that travels with a gpurun snapshot.
# The reference is NOT part of this repository:  `tools/ref_scratch.sh make` before the call, `tools/ref_scratch.sh rm`
# right after it.  Python sources only, no checkpoints / images / notebooks.
set -e
cd "$(dirname "$0")/.."
REF=${DPM_REFERENCE_SRC:-/root/reference}
case "$1" in
  make)
    rm -rf _refscratch; mkdir -p _refscratch
    cp "$REF/dpm_solver_pytorch.py" _refscratch/
    ( cd "$REF" && find examples/stable-diffusion/ldm/models/diffusion/dpm_solver examples/score_sde_pytorch/sampling.py \
        examples/score_sde_pytorch/sde_lib.py examples/score_sde_pytorch/dpm_solver.py examples/score_sde_pytorch/models/__init__.py \
        examples/score_sde_pytorch/models/utils.py examples/ddpm_and_guided-diffusion -name "*.py" -not -path "*__pycache__*" ) \
      | while read -r f; do mkdir -p "_refscratch/$(dirname "$f")"; cp "$REF/$f" "_refscratch/$f"; done
    du -sh _refscratch ;;
  rm) rm -rf _refscratch ;;
  *) echo "usage: $0 make|rm"; exit 2 ;;
esac
