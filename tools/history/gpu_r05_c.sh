#!/bin/bash
# round 5, call C: the elected-reducer experiment of the clustered thresholding kernel -- bit
# equality under forced faults, then the same-box A/B: stage_bench rows + memory-side counters and rocprofv3 rows inside a
# conv-network loop, default protocol vs DPM_TUNE_THR_ELECT = 1, alternating; auto_capture test
TAG=${1:-r05c}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
LAB=$PWD/tools/_variants/lab/libdpm_lab.so
( time DPM_SOLVER_AMD_LIB=$LAB timeout 900 python -m pytest tests -m "gpu and lab" -q -x -k "elected or lds_dma" ) > $O/pytest_lab.log 2>&1; echo "lab tests rc=$?"; tail -3 $O/pytest_lab.log
timeout 600 python -m pytest tests -m gpu -q -x -k "auto_capture or double_precision" > $O/pytest_new.log 2>&1; echo "new product tests rc=$?"; tail -2 $O/pytest_new.log
K='stage_thresh_kernel<float, float, 1, 0, false, 512, 1'
for REP in 1 2; do
  for E in 0 1; do
    DPM_LAB_TUNE="thr_elect=$E" timeout 500 bash tools/profile_stage.sh ${TAG}_b32_e${E}_$REP "cfg5 2M++ thr B=32" "$K" "thr_elect=$E" > $O/prof_b32_e${E}_$REP.log 2>&1; echo "profile B=32 elect=$E rep $REP rc=$?"
    cp gpurun_out/prof_${TAG}_b32_e${E}_$REP/summary.md $O/summary_b32_e${E}_$REP.md 2>/dev/null
    rm -rf gpurun_out/prof_${TAG}_b32_e${E}_$REP
    grep -E "TWO - thr" $O/prof_b32_e${E}_$REP.log | head -3
  done
done
for E in 0 1 0 1; do
  DPM_LAB_TUNE="thr_elect=$E" timeout 300 python tools/stage_bench.py --only "thr" --md $O/stage_thr_e${E}_$RANDOM.md > /dev/null 2>&1; echo "stage_bench thr elect=$E rc=$?"
done
grep -h "TWO - thr +m" $O/stage_thr_e0_*.md | head; echo ---; grep -h "TWO - thr +m" $O/stage_thr_e1_*.md | head
for REP in 1 2; do
  for E in 0 1; do
    DPM_LAB_TUNE="thr_elect=$E" timeout 400 rocprofv3 --kernel-trace --output-format rocpd -d $O/kt_cfg5_e${E}_$REP -o kt -- python tools/in_loop.py --case cfg5 --trajectories 6 > $O/loop_cfg5_e${E}_$REP.log 2>&1; echo "in-loop cfg5 elect=$E rep $REP rc=$?"
    python tools/in_loop.py --summarise $O/kt_cfg5_e${E}_$REP --pattern stage_thresh --md $O/in_loop_cfg5_e${E}_$REP.md > /dev/null 2>&1
    rm -rf $O/kt_cfg5_e${E}_$REP
    sed -n 5,8p $O/in_loop_cfg5_e${E}_$REP.md
  done
done
