#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 900 bash tools/profile_round.sh r02b fp16 > $O/profile_fp16.log 2>&1; echo "profile rc=$?"; tail -45 $O/profile_fp16.log
timeout 600 python bench.py --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "bench32 rc=$?"
timeout 900 python tools/stage_bench.py --md $O/stage_table.md > $O/stage_bench.log 2>&1; echo "stage_bench rc=$?"; tail -50 $O/stage_bench.log
