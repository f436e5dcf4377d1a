#!/bin/bash
# rocprofv3 view of one scenario of tools/stage_bench.py: kernel trace + stats and the memory-side byte counters (separate
# --pmc passes), condensed by tools/rocprof_summary.py.  Launches back to back and behind a 768 MiB eviction sweep
# alternate in that tool, and the profiler adds a few microseconds per launch.
#   usage (through gpurun):  bash tools/profile_stage.sh <tag> <scenario substring> <kernel-name substring> <note>
#   e.g.  bash tools/profile_stage.sh r02_thr "cfg5 2M++ thr B=1024" "stage_thresh_kernel<float, float, 1, 0, false, 512, 1" "..."
set -u
TAG=${1:-r02}
ONLY=${2:-cfg5 2M++ thr B=1024}
KERN=${3:-stage_thresh_kernel<float, float, 1, 0, false, 512, 1}
NOTE=${4:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
export TMPDIR=/tmp
P=$ROOT/gpurun_out/prof_${TAG}
rm -rf "$P"; mkdir -p "$P"
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d "$P/kt" -o kt -- python tools/stage_bench.py --only "$ONLY" > "$P/kt.log" 2> "$P/kt.err"
find "$P/kt" -name "*kernel_stats.csv" -exec cp {} "$P/kernel_stats.csv" \;
find "$P/kt" -name "*kernel_trace.csv" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d "$P/pmc_$C" -o pmc -- python tools/stage_bench.py --only "$ONLY" > "$P/$C.log" 2> "$P/$C.err"
done
python tools/rocprof_summary.py "$P" "$KERN" "$P/summary.md" "$TAG: rocprofv3 ... -- python tools/stage_bench.py --only '$ONLY'  $NOTE" > /dev/null
grep "^| " "$P/kt.log" | head -8
tail -25 "$P/summary.md"
find "$P" -name "*.db" -size +20M -delete
du -sh "$P"
