#!/bin/bash
# rocprofv3 view of the thresholding stage kernel: kernel trace + stats and the memory-side byte counters (separate --pmc
# passes) over `tools/stage_bench.py --only "thr B=1024"` ([1024,3,64,64] fp32, 2M++: one workgroup per sample; launches
# back to back and behind a 768 MiB eviction sweep alternate), condensed by tools/rocprof_summary.py.
#   usage (through gpurun):  bash tools/profile_thresholding.sh <tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
export TMPDIR=/tmp
P=$ROOT/gpurun_out/prof_${TAG}_thr
rm -rf "$P"; mkdir -p "$P"
rocprofv3 --kernel-trace --stats -d "$P/kt" -o kt -- python tools/stage_bench.py --only "cfg5 2M++ thr B=1024" > "$P/kt.log" 2> "$P/kt.err"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d "$P/pmc_$C" -o pmc -- python tools/stage_bench.py --only "cfg5 2M++ thr B=1024" > "$P/$C.log" 2> "$P/$C.err"
done
python tools/rocprof_summary.py "$P" 'stage_thresh_kernel<float, float, 1, 0, false, 512, 1' "$P/summary.md" "$TAG: rocprofv3 ... -- python tools/stage_bench.py --only 'cfg5 2M++ thr B=1024'  ([1024,3,64,64] fp32, 2M++ with dynamic thresholding, k = 1; 5N = 251.66 MB algorithmic bytes per launch)" > /dev/null
grep "thr B=1024" "$P/kt.log" | head -4
tail -25 "$P/summary.md"
find "$P" -name "*.db" -size +20M -delete
du -sh "$P"
