#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel trace + stats, and the memory-side counters in separate
# --pmc passes (never combined with tracing), condensed by tools/rocprof_summary.py.
#   usage (through gpurun):  bash tools/profile_round.sh <tag> [fp16|fp32|bf16]
set -u
TAG=${1:-r02}
DT=${2:-fp16}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
export TMPDIR=/tmp
P=$ROOT/gpurun_out/prof_${TAG}_$DT
rm -rf "$P"; mkdir -p "$P"
python bench.py --dtype $DT > "$P/bench.json" 2> "$P/bench.err"
CMD="python bench.py --dtype $DT --steps 12 --warmup 2 --trajectories-per-step 2 --min-region-s 0.05 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d "$P/kt" -o kt -- $CMD > "$P/bench_kt.json" 2> "$P/bench_kt.err"
find "$P/kt" -name "*kernel_stats.csv" -exec cp {} "$P/kernel_stats.csv" \;
find "$P/kt" -name "*kernel_trace.csv" -delete
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_RDREQ_32B_sum; do
  rocprofv3 --pmc $C -d "$P/pmc_$C" -o pmc -- $CMD > "$P/bench_$C.json" 2> "$P/bench_$C.err"
done
KERN='stage_kernel_multi<__half, __half, 1'
[ "$DT" = fp32 ] && KERN='stage_kernel_multi<float, float, 1'
[ "$DT" = bf16 ] && KERN='stage_kernel_multi<(anonymous namespace)::bf16_t, (anonymous namespace)::bf16_t, 1'
python tools/rocprof_summary.py "$P" "$KERN" "$P/summary.md" "$TAG: rocprofv3 ... -- $CMD  (32 requests of [256,4,64,64] $DT in flight, one fused launch per stage)" > /dev/null
cat "$P/bench.json" | head -c 600; echo; tail -40 "$P/summary.md"
# keep the pulled files small: the databases stay on the box
find "$P" -name "*.db" -size +20M -delete
du -sh "$P"
