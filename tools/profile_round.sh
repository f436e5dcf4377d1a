#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel trace + stats, FETCH_SIZE / WRITE_SIZE in separate
# --pmc passes (never combined with tracing), condensed by tools/rocprof_summary.py.
#   usage (through gpurun):  bash tools/profile_round.sh <tag> [fp16|fp32]
set -u
TAG=${1:-r01}
DT=${2:-fp16}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
export TMPDIR=/tmp
P=$ROOT/gpurun_out/prof_$TAG
rm -rf "$P"; mkdir -p "$P"
python bench.py --dtype $DT > "$P/bench.json" 2> "$P/bench.err"
rocprofv3 --kernel-trace --stats -d "$P/kt" -o kt -- python bench.py --dtype $DT --steps 100 --warmup 10 --no-cpu-baseline > "$P/bench_kt.json" 2> "$P/bench_kt.err"
rocprofv3 --pmc FETCH_SIZE -d "$P/fetch" -o fetch -- python bench.py --dtype $DT --steps 20 --warmup 2 --no-cpu-baseline > "$P/bench_fetch.json" 2> "$P/bench_fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$P/write" -o write -- python bench.py --dtype $DT --steps 20 --warmup 2 --no-cpu-baseline > "$P/bench_write.json" 2> "$P/bench_write.err"
KERN='stage_kernel<__half, __half, 1'
[ "$DT" = fp32 ] && KERN='stage_kernel<float, float, 1'
python tools/rocprof_summary.py "$P" "$KERN" "$P/summary.md" "$TAG: rocprofv3 ... -- python bench.py --dtype $DT --steps 100 --warmup 10 --no-cpu-baseline ([256,4,64,64], 8 buffer sets, eager native loop)" > /dev/null
cat "$P/bench.json"; tail -30 "$P/summary.md"
du -sh "$P"
