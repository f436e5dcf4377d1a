cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06q
( timeout 600 python -m pytest tests/test_gpu_extensions.py -m gpu -q -x -k "classifier_gradient or fuzz_slice_on or adaptive_host_loop" ) > gpurun_out/r06q/newtests.log 2>&1; echo "newtests rc=$?"; tail -5 gpurun_out/r06q/newtests.log
for SEED in 1 2 3 4; do
  timeout 300 python tools/fuzz_gpu.py --cases 2000 --seed $SEED --case-timeout 20 --out gpurun_out/r06q/fuzz_gpu_seed$SEED.json > gpurun_out/r06q/fuzz_gpu_seed$SEED.log 2>&1; echo "seed $SEED rc=$?"
  tail -1 gpurun_out/r06q/fuzz_gpu_seed$SEED.log | cut -c1-420; grep -A3 "^case\|Timeout" gpurun_out/r06q/fuzz_gpu_seed$SEED.log | cut -c1-600 | head -30; cat gpurun_out/r06q/*current_case.txt 2>/dev/null | cut -c1-600
done
