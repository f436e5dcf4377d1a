#!/bin/bash
# round 5, call D: what the round's profiles/ quote -- the full GPU suite, smoke, the default bench line (+ the driver's N = 1
# torchrun line), rocprofv3 kernel stats and memory-side counters of the headline command, the stage kernel inside the conv
# loop by rows (fp16 / fp32), cold start of the trimmed library, the other dtype lines
TAG=${1:-r05d}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|skipped|real" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc=$?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun_1rank.err; echo "torchrun 1 rank rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_plain_nosec.json 2> $O/bench_plain_nosec.err; echo "plain no-secondary rc=$?"
python - <<PY
import json
for f in ("bench_default","bench_torchrun_1rank","bench_plain_nosec"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["value"], d["value_per_gpu"], d["roofline"]["frac"], d["launcher"], d["steps"])
d=json.load(open("$O/bench_default.json")); r=d["roofline"]; inl=r.get("in_network_loop",{})
print("in_loop", {k: inl.get(k) for k in ("stage_kernel_us","frac","frac_of_floor","stage_added_wall_us","stage_added_wall_iqr_us","error")})
print("floor", inl.get("floor")); print("cold", r.get("single_request_cold",{}).get("kernel_us"), "cache_resident", r.get("cache_resident",{}).get("frac"))
PY
timeout 900 bash tools/profile_round.sh $TAG fp16 > $O/profile_fp16.log 2>&1; echo "profile rc=$?"; tail -30 $O/profile_fp16.log
for DT in fp16 fp32; do
  timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_loopc_$DT -o kt -- python tools/in_loop.py --dtype $DT --kinds conv --trace-only > $O/kt_loopc_$DT.log 2>&1; echo "rocprof in-loop conv $DT rc=$?"
  python tools/in_loop.py --summarise $O/kt_loopc_$DT --md $O/in_loop_trace_conv_$DT.md --title "stage kernel inside a torch network loop, conv network (rocprofv3 --kernel-trace)" > /dev/null 2>&1
  find $O/kt_loopc_$DT -name "*kernel_stats.csv" -exec cp {} $O/in_loop_conv_kernel_stats_$DT.csv \;
  rm -rf $O/kt_loopc_$DT
  sed -n 5,9p $O/in_loop_trace_conv_$DT.md
done
( time timeout 600 python tools/cold_start.py --repeat 3 --out $O/cold_start.json ) > $O/cold_start.log 2>&1; echo "cold start rc=$?"; grep cold_start_ms $O/cold_start.log | cut -c1-260
timeout 600 python bench.py --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "bench fp32 rc=$?"
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc=$?"
timeout 600 python bench.py --dtype fp32 --eps-dtype fp16 --no-cpu-baseline > $O/bench_fp32_fp16.json 2> $O/bench_fp32_fp16.err; echo "bench fp32/fp16 rc=$?"
python - <<PY
import json
for f in ("bench_fp32","bench_bf16","bench_fp32_fp16"):
    d=json.load(open("$O/%s.json"%f)); inl=d["roofline"].get("in_network_loop",{}); print(f, d["value"], d["roofline"]["frac"], inl.get("stage_kernel_us"), inl.get("frac"), inl.get("frac_of_floor"))
PY
