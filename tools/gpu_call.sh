#!/bin/bash
# One parameterised GPU call (replaces the per-call scripts of rounds 1-5, now under tools/history/).
#   usage (through gpurun):  bash tools/gpu_call.sh <tag> <step> [<step> ...]
# Every step writes under gpurun_out/<tag>/ and prints one summary line; a failing step does not stop the later ones.
#   tests        the GPU suite (pytest -m gpu), timed
#   smoke        __graft_entry__.smoke()
#   bench        the default bench line (what the driver runs), timed
#   guard5       tests/test_zz_perf_guard.py five times (its record: perf_guard.jsonl)
#   torchrun1    the driver's N > 1 launch line with one rank (RCCL communicator, preflight, all-gather) + --preflight alone
#   configs      tools/config_bench.py: figures + rocprofv3 kernel rows per BASELINE configuration (frozen model_fn, product library)
#   configsloop  the same cases inside a conv-network loop, kernel-only (lab build)
#   profile      tools/profile_round.sh (headline: kernel trace + counters)
#   cpubase      the unmodified reference + the numpy port on this box's host cores (needs _refscratch/: tools/ref_scratch.sh make)
#   dropin       the reference's own example call sites on the engine (needs _refscratch/)
#   bigsingle    tools/big_single.py: launch shape of ONE large stage launch (lab build)
#   newtests     the GPU tests added this round (quick iteration before the full suite)
#   fuzzapi      tools/fuzz_gpu_api.py: the same for the GPU-side extensions (requests, capture, auto_capture, NHWC, streams, half states, MaskBlend, device adaptive)
#   fuzzcapi     tools/fuzz_gpu_capi.py: the C ABI's native sample loop (dpm_plan_run / _multi / dpm_graph_*) with a model callback vs sample()
#   fuzzmethods  tools/fuzz_gpu_methods.py: the public per-update / evaluation / schedule methods on the GPU vs the double
#   fuzzkernel   tools/fuzz_gpu_kernel.py: one dpm_stage_launch per case (random stage record, tiling-boundary geometry, dtype pairs, layouts) vs the double
#   fuzzthresh   tools/fuzz_gpu_thresh.py: dynamic_thresholding_fn on the GPU vs torch.quantile on the CPU (boundary sample sizes, ties, any ratio)
#   sweeps       test_thresholded_sampling_random_sweep extended: 10 000 configurations on the product, 3 x 3000 under forced faults + 3000 on the general route (lab build)
#   torchprobe   tools/torch_on_device_probe.py: torch's own half * scalar and quantile interpolation on this GPU vs its CPU kernels
#   fuzzgpu      tools/fuzz_gpu.py: the drop-in fuzz's random cases, engine on the GPU vs the engine's host code on the numpy double
TAG=${1:?tag}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
[ -f _refscratch/dpm_solver_pytorch.py ] && export DPM_REFERENCE_DIR=_refscratch
for STEP in "$@"; do
case $STEP in
tests)
  ( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"
  grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest.log | head; grep real $O/pytest.log ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log ;;
bench)
  ( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc=$?"; tail -3 $O/bench_default.err
  python - <<PY
import json
d = json.load(open("$O/bench_default.json")); r = d["roofline"]
print("value", d["value"], "frac", r["frac"], "ms_per_step", d["ms_per_step"], "lone cold", r.get("single_request_cold_us"), r.get("single_request_cold_frac"))
print("preflight", d.get("preflight", {}).get("ok"), d.get("preflight", {}).get("seconds"), "ordinals", d.get("device_ordinals"))
for k, v in d.get("configs", {}).items():
    print(" ", k, v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "measured_in_this_run"})
print("configs seconds", d.get("configs_detail", {}).get("seconds"), "cpu_baseline", {k: v for k, v in d.get("cpu_baseline", {}).items() if not isinstance(v, (dict, list)) and k != "sample"})
print("line bytes", len(json.dumps(d)), "configs bytes", len(json.dumps(d.get("configs", {}))))
PY
  ;;
guard5)
  rm -f gpurun_out/perf_guard.jsonl
  for i in 1 2 3 4 5; do
    ( time timeout 300 python -m pytest tests/test_zz_perf_guard.py -m gpu -q -s -p no:cacheprovider ) > $O/guard_$i.log 2>&1; echo "guard $i rc=$? $(grep -E 'passed|failed' $O/guard_$i.log | tail -1) $(grep real $O/guard_$i.log)"
  done
  cp gpurun_out/perf_guard.jsonl $O/perf_guard.jsonl 2>/dev/null; cat $O/perf_guard.jsonl ;;
torchrun1)
  ( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun_1rank.err ); echo "torchrun 1 rank rc=$?"
  grep preflight $O/bench_torchrun_1rank.err
  ( time timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --preflight > $O/preflight_1rank.json 2> $O/preflight_1rank.err ); echo "preflight-only rc=$?"; cat $O/preflight_1rank.json | cut -c1-600 ;;
configs)
  timeout 600 python tools/config_bench.py --out $O/configs.json > $O/configs.log 2>&1; echo "config_bench rc=$?"; cut -c1-420 $O/configs.log
  for CASE in cfg1 cfg3 cfg5 cfg_sd64 one8192; do
    timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_$CASE -o kt -- python tools/config_bench.py --cases $CASE --trace-only > $O/kt_$CASE.log 2>&1; echo "rocprof $CASE rc=$?"
    find $O/kt_$CASE -name "*kernel_stats.csv" -exec cp {} $O/configs_${CASE}_kernel_stats.csv \;
    rm -rf $O/kt_$CASE
    head -4 $O/configs_${CASE}_kernel_stats.csv | cut -c1-260
  done ;;
configsloop)
  timeout 600 python tools/config_bench.py --in-loop --cases cfg1,cfg3,cfg5,cfg_sd64 --out $O/configs_in_loop.json > $O/configs_in_loop.log 2>&1; echo "config_bench in-loop rc=$?"; cut -c1-420 $O/configs_in_loop.log ;;
profile)
  timeout 900 bash tools/profile_round.sh $TAG fp16 > $O/profile_fp16.log 2>&1; echo "profile rc=$?"; tail -32 $O/profile_fp16.log ;;
cpubase)
  if [ -f _refscratch/dpm_solver_pytorch.py ]; then
    timeout 500 python tools/cpu_baseline.py --budget 40 --out $O/cpu_baseline_reference_gpubox.json --where "MI355X box host cores (gpurun)" > $O/cpu_baseline.log 2>&1; echo "cpu_baseline rc=$?"
    python -c "import json; d=json.load(open('$O/cpu_baseline_reference_gpubox.json')); print(d['value'], d['cores'], d.get('port_over_reference'))"
  else echo "cpubase: no _refscratch/"; fi ;;
dropin)
  if [ -d _refscratch/examples ]; then
    timeout 300 python tools/dropin_examples.py --device cuda:0 --out $O/dropin.json > $O/dropin.log 2>&1; echo "dropin rc=$?"; tail -4 $O/dropin.log
  else echo "dropin: no _refscratch/"; fi ;;
bigsingle)
  timeout 900 python tools/big_single.py --out $O/big_single.json > $O/big_single.log 2>&1; echo "big_single rc=$?"; grep '^{' $O/big_single.log | cut -c1-330; grep -v '^{' $O/big_single.log | tail -5 ;;
newtests)
  ( time timeout 900 python -m pytest tests -m gpu -q -x -k "${NEWTESTS_K:-thresh}" ) > $O/newtests.log 2>&1; echo "newtests rc=$?"; tail -15 $O/newtests.log ;;
fuzzgpu)
  for SEED in ${FUZZ_SEEDS:-0 1}; do
    timeout 1500 python tools/fuzz_gpu.py --cases ${FUZZ_CASES:-1500} --seed $SEED --out $O/fuzz_gpu_seed$SEED.json > $O/fuzz_gpu_seed$SEED.log 2>&1; echo "fuzz_gpu seed $SEED rc=$?"
    tail -1 $O/fuzz_gpu_seed$SEED.log | cut -c1-700; grep -c "^case" $O/fuzz_gpu_seed$SEED.log
  done ;;
fuzzapi)
  for SEED in ${FUZZ_SEEDS:-0 1}; do
    timeout 1200 python tools/fuzz_gpu_api.py --cases ${FUZZ_CASES:-1200} --seed $SEED --case-timeout 30 --out $O/fuzz_gpu_api_seed$SEED.json > $O/fuzz_gpu_api_seed$SEED.log 2>&1; echo "fuzz_gpu_api seed $SEED rc=$?"
    tail -1 $O/fuzz_gpu_api_seed$SEED.log | cut -c1-1500; grep -c "^case" $O/fuzz_gpu_api_seed$SEED.log; cat $O/*current_case.txt 2>/dev/null | cut -c1-700
  done ;;
fuzzcapi)
  for SEED in ${FUZZ_SEEDS:-0 1}; do
    timeout 1200 python tools/fuzz_gpu_capi.py --cases ${FUZZ_CASES:-1500} --seed $SEED --case-timeout 30 --out $O/fuzz_gpu_capi_seed$SEED.json > $O/fuzz_gpu_capi_seed$SEED.log 2>&1; echo "fuzz_gpu_capi seed $SEED rc=$?"
    tail -1 $O/fuzz_gpu_capi_seed$SEED.log | cut -c1-900; grep -c "^case" $O/fuzz_gpu_capi_seed$SEED.log; grep -A2 "^case" $O/fuzz_gpu_capi_seed$SEED.log | cut -c1-700 | head -24; cat $O/*current_case.txt 2>/dev/null | cut -c1-700
  done ;;
fuzzmethods)
  for SEED in ${FUZZ_SEEDS:-0 1}; do
    timeout 1200 python tools/fuzz_gpu_methods.py --cases ${FUZZ_CASES:-4000} --seed $SEED --case-timeout 30 --out $O/fuzz_gpu_methods_seed$SEED.json > $O/fuzz_gpu_methods_seed$SEED.log 2>&1; echo "fuzz_gpu_methods seed $SEED rc=$?"
    tail -1 $O/fuzz_gpu_methods_seed$SEED.log | cut -c1-500; grep -c "^call" $O/fuzz_gpu_methods_seed$SEED.log; grep -A2 "^call" $O/fuzz_gpu_methods_seed$SEED.log | cut -c1-600 | head -30; cat $O/*current_case.txt 2>/dev/null | cut -c1-700
  done ;;
fuzzkernel)
  for SEED in ${FUZZ_SEEDS:-0 1}; do
    timeout 1400 python tools/fuzz_gpu_kernel.py ${FUZZ_KERNEL_FLAGS:-} --cases ${FUZZ_CASES:-3000} --seed $SEED --case-timeout 60 --out $O/fuzz_gpu_kernel_seed$SEED.json > $O/fuzz_gpu_kernel_seed$SEED.log 2>&1; echo "fuzz_gpu_kernel seed $SEED rc=$?"
    tail -1 $O/fuzz_gpu_kernel_seed$SEED.log | cut -c1-1500; grep -c "^case" $O/fuzz_gpu_kernel_seed$SEED.log; grep -A2 "^case" $O/fuzz_gpu_kernel_seed$SEED.log | cut -c1-600 | head -40; cat $O/*current_case.txt 2>/dev/null | cut -c1-700
  done ;;
fuzzthresh)
  for SEED in ${FUZZ_SEEDS:-0 1}; do
    timeout 1400 python tools/fuzz_gpu_thresh.py --cases ${FUZZ_CASES:-3000} --seed $SEED --case-timeout 60 --out $O/fuzz_gpu_thresh_seed$SEED.json > $O/fuzz_gpu_thresh_seed$SEED.log 2>&1; echo "fuzz_gpu_thresh seed $SEED rc=$?"
    tail -1 $O/fuzz_gpu_thresh_seed$SEED.log | cut -c1-1500; grep -c "^case" $O/fuzz_gpu_thresh_seed$SEED.log; grep -A2 "^case" $O/fuzz_gpu_thresh_seed$SEED.log | cut -c1-600 | head -40; cat $O/*current_case.txt 2>/dev/null | cut -c1-700
  done ;;
sweeps)
  # the long thresholded-sampling sweeps: the product (DPM_THR_SWEEP configurations), then the lab build under forced faults
  ( time DPM_THR_SWEEP=${SWEEP_N:-10000} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k test_thresholded_sampling_random_sweep ) > $O/sweep_product.log 2>&1; echo "sweep product rc=$? $(grep -E 'passed|failed' $O/sweep_product.log | tail -1) $(grep real $O/sweep_product.log)"
  for F in 1 2 3; do
    ( time DPM_SOLVER_AMD_LIB=tools/_variants/lab/libdpm_lab.so DPM_THR_SWEEP=${SWEEP_FAULT_N:-3000} DPM_THR_SWEEP_FAULT=$F timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k test_thresholded_sampling_random_sweep ) > $O/sweep_fault$F.log 2>&1; echo "sweep fault $F rc=$? $(grep -E 'passed|failed' $O/sweep_fault$F.log | tail -1) $(grep real $O/sweep_fault$F.log)"
  done
  ( time DPM_SOLVER_AMD_LIB=tools/_variants/lab/libdpm_lab.so DPM_THR_SWEEP=${SWEEP_FAULT_N:-3000} DPM_THR_SWEEP_ONE_HOP=0 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k test_thresholded_sampling_random_sweep ) > $O/sweep_general.log 2>&1; echo "sweep general route rc=$? $(grep -E 'passed|failed' $O/sweep_general.log | tail -1) $(grep real $O/sweep_general.log)" ;;
torchprobe)
  timeout 600 python tools/torch_on_device_probe.py > $O/torch_on_device.json 2> $O/torch_on_device.err; echo "torchprobe rc=$?"; cat $O/torch_on_device.json ;;
*) echo "unknown step $STEP" ;;
esac
done
find $O -name "*.db" -size +20M -delete
du -sh $O
