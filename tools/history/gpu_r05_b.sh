#!/bin/bash
# round 5, call B: full GPU suite (product + lab subprocess; double precision, LDS-DMA), the lone launch by rocprofv3 rows --
# stage kernel (LDS-DMA / register path) and the best floor configurations in the same slot --, bench
TAG=${1:-r05b}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|skipped|real" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20
CFG="p1_r1_b512_g16_nt1_pr0_st0,p1_r1_b512_g0_nt1_pr0_st0,p1_r1_b1024_g8_nt1_pr0_st0,p1_r2_b256_g0_nt1_pr0_st0,p0_r1_b512_g0_nt1_pr0_st0,p0_r1_b1024_g8_nt1_pr0_st0,p0_r1_b256_g8_nt1_pr0_st0,p0_r2_b256_g0_nt1_pr0_st0"
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/kt_floor -o kt -- python tools/floor.py --trace-only --trajectories 4 --configs $CFG --seq $O/floor_seq.json > $O/floor_trace.log 2>&1; echo "floor trace rc=$?"; tail -2 $O/floor_trace.log
python tools/floor.py --summarise $O/kt_floor --seq $O/floor_seq.json --out $O/floor_rows.json > $O/floor_rows.log 2>&1; echo "floor rows rc=$?"; tail -60 $O/floor_rows.log
rm -rf $O/kt_floor
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
r=d["roofline"]
print("value", d["value"], "frac", r["frac"], "launch_us", r["launch_us"])
print("in_loop", {k: r["in_network_loop"].get(k) for k in ("stage_kernel_us","frac","frac_of_floor","stage_added_wall_us","error")})
print("floor", r["in_network_loop"].get("floor"))
print("single_request_cold", r.get("single_request_cold",{}).get("kernel_us"), "nac", r.get("no_arithmetic_ceiling"))
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value","kind","cores","measured_in_this_run")})
PY
tail -3 $O/bench_default.err
