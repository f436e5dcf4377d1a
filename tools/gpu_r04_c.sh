#!/bin/bash
# round 4, call c: the other BASELINE kernels inside a real torch network loop (rocprofv3 rows), channels_last vs default
# layout, and the thresholding kernel after the fault-recovery change (routes, stage rows, traffic counters at cfg5's size)
TAG=${1:-r04c}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for CASE in nchw nhwc cfg_sd64 cfg_sd8 cfg5 cfg3; do
  TR=6; [ $CASE = cfg3 ] && TR=3
  timeout 420 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/kt_$CASE -o kt -- python tools/in_loop.py --case $CASE --trajectories $TR > $O/case_$CASE.log 2>&1; echo "rocprof $CASE rc=$?"
  grep '^{' $O/case_$CASE.log | tail -1
  python tools/in_loop.py --summarise $O/kt_$CASE --md $O/in_loop_$CASE.md --title "BASELINE kernel inside a torch network loop: case $CASE (rocprofv3 --kernel-trace --stats -- python tools/in_loop.py --case $CASE)" > /dev/null 2>&1
  find $O/kt_$CASE -name "*kernel_stats.csv" -exec cp {} $O/in_loop_${CASE}_kernel_stats.csv \;
  rm -rf $O/kt_$CASE
  sed -n 3,12p $O/in_loop_$CASE.md | cut -c1-220
done
timeout 600 python tools/thr_routes.py > $O/thr_routes.txt 2>&1; echo "routes rc=$?"; grep -A6 "shape (32" $O/thr_routes.txt | head -16; grep -A4 "shape (64" $O/thr_routes.txt | head -12
timeout 600 python tools/stage_bench.py --only "thr" --md $O/stage_thr.md > $O/stage_thr.log 2>&1; echo "stage_bench thr rc=$?"; grep -i "thr" $O/stage_thr.md | head -14
# cfg5's own size: kernel rows + memory-side traffic (separate --pmc passes)
P=$O/prof_thr32; mkdir -p $P
CMD="python tools/stage_bench.py --only cfg5"
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $P/kt -o kt -- $CMD > $P/kt.log 2>&1; echo "rocprof thr32 kt rc=$?"
find $P/kt -name "*kernel_stats.csv" -exec cp {} $P/kernel_stats.csv \;
find $P/kt -name "*kernel_trace.csv" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $P/pmc_$C -o pmc -- $CMD > $P/pmc_$C.log 2>&1; echo "pmc $C rc=$?"
done
python tools/rocprof_summary.py $P "stage_thresh_kernel<float, float, 1, 0, false, 512, 1" $P/summary.md "$TAG: rocprofv3 ... -- $CMD (cfg5's own size [32,3,64,64])" > /dev/null 2>&1
tail -12 $P/summary.md
find $P -name "*.db" -size +20M -delete
# two processes on one GPU: what the workers report
D=$(mktemp -d); for R in 0 1; do python tests/cluster_pair_worker.py $D $R 2 > $O/pair_$R.json 2> $O/pair_$R.err & done; wait; cat $O/pair_0.json $O/pair_1.json
du -sh $O
# EXPERIMENT: resident stage kernel vs dispatched (conv network), fp16
timeout 600 python tools/in_loop.py --kinds conv --dtype fp16 --resident 256:1,512:1,1024:1,2048:1,1024:4,1024:0 --trajectories 8 --out $O/resident_fp16.json > $O/resident_fp16.log 2>&1; echo "resident rc=$?"; grep '^{' $O/resident_fp16.log | cut -c1-400; tail -3 $O/resident_fp16.log | cut -c1-300
