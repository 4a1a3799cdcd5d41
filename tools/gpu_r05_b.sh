#!/bin/bash
# round 5, GPU call B: the suite on the final stage plan (1+2+3+4+4+4+2) with the error-bound assertions, the bench lines,
# the real-topology graph with and without the locality numbering, the N > 1 code path at world 1 (oracle probe, strong
# legs), rocprofv3 kernel stats + PMC passes of the cfg 3 command
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05b}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
( time timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 --tb=short --durations=8 ) > "$OUT/gpu_tests.log" 2>&1
tail -22 "$OUT/gpu_tests.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
python - "$TAG" <<'P'
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}/bench_cfg3.json"))
print("cfg3:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms_by_mode"], d["roofline"]["traffic"], d["parity_spot_check"]["max_rel_score_err"], d["ppr_contract"]["ppr_residual_max"], d["ppr_contract"]["meets_default_tol"], "contract", d["with_convergence_contract"].get("value"), d["with_convergence_contract"].get("sweeps_used_max"), "accel", d["with_accelerated_stages"].get("value"), d["with_accelerated_stages"].get("with_convergence_contract",{}).get("value"))
P
for LOC in none auto; do
  timeout 600 python bench.py --config real2wiki --locality $LOC --steps 20 --warmup 5 --cpu-queries 6 --cpu-budget-s 15 --cpu-vec-queries 8 > "$OUT/bench_real2wiki_$LOC.json" 2> "$OUT/bench_real2wiki_$LOC.err"
  python - "$TAG" "$LOC" <<'P'
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}/bench_real2wiki_{sys.argv[2]}.json"))
print("real2wiki", sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms_by_mode"], d["config"]["locality_score_after_renumbering"], d["config"]["engine_opt_flags"], d["parity_spot_check"]["max_rel_score_err"], d["parity_spot_check"]["exact_id_fraction"])
P
done
timeout 300 python bench.py --config real2wiki1 --locality auto --steps 50 --warmup 5 --cpu-queries 8 --cpu-budget-s 10 --cpu-vec-queries 8 > "$OUT/bench_real2wiki1_auto.json" 2> "$OUT/bench_real2wiki1_auto.err"; tail -c 300 "$OUT/bench_real2wiki1_auto.json"; echo
HRAG_FORCE_DIST=1 HRAG_STRONG_GLOBAL_BATCH=128 timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > "$OUT/bench_dist_world1.json" 2> "$OUT/bench_dist_world1.err"
python - "$TAG" <<'P'
import json,sys
try:
    d=json.load(open(f"gpurun_out/{sys.argv[1]}/bench_dist_world1.json"))
    print("world1:", d["value"], d["value_leg"], d["value_rowshard"], d["value_hybrid"], d["value_replica"], d["rowshard"]["parity"], d["hybrid"]["parity"], d.get("configs3_strong",{}).get("value"), d.get("configs3_strong",{}).get("value_leg"))
except Exception as e:
    print("world1 FAILED", e); print(open(f"gpurun_out/{sys.argv[1]}/bench_dist_world1.err").read()[-1500:])
P
HRAG_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --collective allreduce --no-cpu-baseline > "$OUT/bench_dist_world1_allreduce.json" 2> "$OUT/bench_dist_world1_allreduce.err"; tail -c 200 "$OUT/bench_dist_world1_allreduce.json"; echo
for CFG in cfg1s cfg2; do
  timeout 600 python bench.py --config $CFG --steps 50 --warmup 5 --cpu-queries 16 > "$OUT/bench_$CFG.json" 2> "$OUT/bench_$CFG.err"
  python - "$TAG" "$CFG" <<'P'
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}/bench_{sys.argv[2]}.json"))
print(sys.argv[2], d["value"], d["roofline"]["frac"], d["parity_spot_check"]["max_rel_score_err"], d["with_accelerated_stages"].get("value"))
P
done
bash tools/gpu_profile.sh "$TAG/prof" --steps 5 --warmup 1 > "$OUT/profile.log" 2>&1
tail -3 "$OUT/profile.log"
python tools/prof_summary.py "$OUT/prof" "$OUT/${TAG}_cfg3" > "$OUT/prof_summary.log" 2>&1; cat "$OUT/prof_summary.log"
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
du -sh "$OUT"
