#!/bin/bash
# round-2 GPU call A: tests, fused-kernel sweep, bench, counter list
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout 900 python tools/tune_multi.py > $O/tune_multi.txt 2> $O/tune_multi.err; echo "tune rc=$?"; tail -3 $O/tune_multi.err
timeout 600 python bench.py > $O/bench_fp16.json 2> $O/bench_fp16.err; echo "bench rc=$?"; tail -3 $O/bench_fp16.err
timeout 600 python bench.py --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "bench32 rc=$?"
(cd /tmp && timeout 120 rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_all.txt 2>&1); grep -i -E "dram|mall|hbm|EA0|EA_|MC_|UMC|DF_|fabric" $O/counters_all.txt | head -150 > $O/counters_mem.txt; wc -l $O/counters_all.txt $O/counters_mem.txt
head -c 1500 $O/bench_fp16.json
