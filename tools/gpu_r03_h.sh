#!/bin/bash
# round 3, call h: 16-byte permuted loads of 2-byte network outputs next to an fp32 state: parity + stage rows + fused bench
TAG=${1:-r03h}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python tools/stage_bench.py --only "f16 eps" --md $O/stage_mixed.md > $O/stage_mixed.log 2>&1; echo "stage_bench mixed rc=$?"; grep -i "f16 eps" $O/stage_mixed.md | head -14
timeout 600 python bench.py --dtype fp32 --eps-dtype fp16 --no-cpu-baseline --loop-net none > $O/bench_fp32_fp16.json 2> $O/bench_fp32_fp16.err; echo "bench fp32/fp16 rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_fp32_fp16.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['frac'], r['launch_us'], r['single_request_cold']['kernel_us'], r['cache_resident']['kernel_only_us'])"
