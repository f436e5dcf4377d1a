#!/bin/bash
# round 3, first GPU call: GPU suite, the reference's own CPU timing on this box's host cores, the stage kernel inside a
# torch network loop (events + rocprofv3 kernel trace, prefetch experiment), the default bench line
TAG=${1:-r03a}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
nproc > $O/nproc.txt; grep -m1 "model name" /proc/cpuinfo >> $O/nproc.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
if [ -f _refscratch/dpm_solver_pytorch.py ]; then
  DPM_REFERENCE_DIR=_refscratch timeout 300 python tools/cpu_baseline.py --out $O/cpu_baseline_reference_gpubox.json --where "MI355X box host cores (gpurun)" > $O/cpu_baseline.log 2>&1; echo "cpu_baseline rc=$?"; tail -5 $O/cpu_baseline.log
fi
timeout 600 python tools/in_loop.py --kinds gemm --out $O/in_loop_gemm.json > $O/in_loop_gemm.log 2>&1; echo "in_loop gemm rc=$?"; tail -4 $O/in_loop_gemm.log
timeout 420 python tools/in_loop.py --kinds conv --out $O/in_loop_conv.json > $O/in_loop_conv.log 2>&1; echo "in_loop conv rc=$?"; tail -4 $O/in_loop_conv.log
for PF in None 0; do
  timeout 420 rocprofv3 --kernel-trace --stats -d $O/kt_pf$PF -o kt -- python tools/in_loop.py --trace-only --prefetch $PF > $O/kt_pf$PF.log 2>&1; echo "rocprof pf=$PF rc=$?"
  python tools/in_loop.py --summarise $O/kt_pf$PF --md $O/in_loop_trace_pf$PF.md > /dev/null 2>&1
  find $O/kt_pf$PF -name "*kernel_stats.csv" -exec cp {} $O/in_loop_kernel_stats_pf$PF.csv \;
  find $O/kt_pf$PF -name "*.db" -size +20M -delete
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; head -c 1500 $O/bench_default.json
du -sh $O
