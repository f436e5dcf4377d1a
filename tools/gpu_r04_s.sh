#!/bin/bash
# round 4, call s: the fused multi-request launch (the bench's kernel) with 512-thread workgroups against 256, alternating
TAG=${1:-r04s}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q -x -k "workgroup_size or multi or fused or requests" > $O/pytest_bt.log 2>&1; echo "bit identity + fused tests rc=$?"; tail -2 $O/pytest_bt.log
for REP in 1 2 3; do
  for BT in 256 512; do
    DPM_BENCH_BLOCK_THREADS=$BT timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_fp16_${BT}_$REP.json 2>$O/bench_fp16_${BT}_$REP.err
    python -c "import json;d=json.loads(open('$O/bench_fp16_${BT}_$REP.json').read().strip().splitlines()[-1]);print('fp16 bt=$BT rep $REP frac',d['roofline']['frac'],'launch_us',d['roofline'].get('launch_us'))"
  done
done
for BT in 256 512; do
  DPM_BENCH_BLOCK_THREADS=$BT timeout 300 python bench.py --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_fp32_${BT}.json 2>$O/bench_fp32_${BT}.err
  python -c "import json;d=json.loads(open('$O/bench_fp32_${BT}.json').read().strip().splitlines()[-1]);print('fp32 bt=$BT frac',d['roofline']['frac'])"
done
