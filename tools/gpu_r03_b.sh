#!/bin/bash
# round 3, second GPU call: reference CPU timing with a thread-count sweep, prefetch experiment v2 (hook before the network's
# last KERNEL, resident kernel variant), the new first-stage mask-blend kernel (tests + stage table rows)
TAG=${1:-r03b}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_extensions.py -m gpu -q -x -k "maskblend" > $O/pytest_blend.log 2>&1; echo "pytest blend rc=$?"; tail -3 $O/pytest_blend.log
if [ -f _refscratch/dpm_solver_pytorch.py ]; then
  DPM_REFERENCE_DIR=_refscratch timeout 400 python tools/cpu_baseline.py --budget 40 --out $O/cpu_baseline_reference_gpubox.json --where "MI355X box host cores (gpurun)" > $O/cpu_baseline.log 2>&1; echo "cpu_baseline rc=$?"; head -12 $O/cpu_baseline.log
fi
timeout 600 python tools/in_loop.py --kinds gemm --out $O/in_loop_gemm.json > $O/in_loop_gemm.log 2>&1; echo "in_loop gemm rc=$?"; grep -o '"stage_kernel_us": [0-9.]*\|"stage_added_wall_us": [0-9.]*\|"network_ms_per_call": [0-9.]*\|"variant": "[^"]*"' $O/in_loop_gemm.log
for PF in None 0; do
  timeout 420 rocprofv3 --kernel-trace --stats -d $O/kt_pf$PF -o kt -- python tools/in_loop.py --trace-only --prefetch $PF > $O/kt_pf$PF.log 2>&1; echo "rocprof pf=$PF rc=$?"
  python tools/in_loop.py --summarise $O/kt_pf$PF --md $O/in_loop_trace_pf$PF.md > /dev/null 2>&1
  find $O/kt_pf$PF -name "*kernel_stats.csv" -exec cp {} $O/in_loop_kernel_stats_pf$PF.csv \;
  find $O/kt_pf$PF -name "*.db" -delete
  head -12 $O/in_loop_trace_pf$PF.md
done
timeout 600 python tools/stage_bench.py --only "inpaint" --md $O/stage_inpaint.md > $O/stage_inpaint.log 2>&1; echo "stage_bench inpaint rc=$?"; grep -i "blend" $O/stage_inpaint.md
timeout 600 python tools/stage_bench.py --only "2M++ cfg, f32" --md $O/stage_cfg.md > $O/stage_cfg.log 2>&1; echo "stage_bench cfg rc=$?"; grep -i "cfg" $O/stage_cfg.md | head
du -sh $O
