#!/bin/bash
# One GPU call with everything a round's profiles/ need: tests, bench lines (fp16 / fp32 / bf16), rocprofv3 kernel trace and
# memory-side counters of the headline command, the stage table.   usage (through gpurun): bash tools/gpu_round.sh r02
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
# the launch line the driver uses for N > 1, with one rank (RCCL communicator, barrier, MAX all-reduce, final all-gather)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun_1rank.err; echo "torchrun 1 rank rc=$?"
timeout 900 bash tools/profile_round.sh $TAG fp16 > $O/profile_fp16.log 2>&1; echo "profile rc=$?"; tail -32 $O/profile_fp16.log
timeout 600 python bench.py --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "bench fp32 rc=$?"
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc=$?"
timeout 600 python bench.py --dtype fp32 --eps-dtype fp16 --no-cpu-baseline > $O/bench_fp32_fp16.json 2> $O/bench_fp32_fp16.err; echo "bench fp32/fp16 rc=$?"
timeout 900 python tools/stage_bench.py --md $O/stage_table.md > $O/stage_bench.log 2>&1; echo "stage_bench rc=$?"; tail -8 $O/stage_bench.log
