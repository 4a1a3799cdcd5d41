#!/bin/bash
# round 5, GPU call A: the whole GPU suite (new full-size / real-topology / soak-slice tests), smoke, the cfg 3 bench line
# on the new stage plan, two stage plans beside it (HRAG_P8_PLAN: the round-4 plan and a 6-stage one), the real-topology graph
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05a
mkdir -p "$OUT"
cd "$REPO"
( time timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 --tb=short --durations=12 ) > "$OUT/gpu_tests.log" 2>&1
tail -25 "$OUT/gpu_tests.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
python - <<'P'
import json
d=json.load(open("gpurun_out/r05a/bench_cfg3.json"))
print("cfg3 new plan:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms_by_mode"], d["parity_spot_check"]["max_rel_score_err"], d["parity_spot_check"]["exact_id_fraction"], d["ppr_contract"]["ppr_residual_max"], d["with_convergence_contract"].get("value"), d["with_accelerated_stages"].get("value"))
P
for PLAN in 1,2,3,3,3,3,3,2 1,3,4,4,5,3 1,2,4,4,4,4,1; do
  HRAG_P8_PLAN=$PLAN timeout 600 python bench.py --steps 20 --warmup 5 --no-accel --ppr-tol 0 --cpu-queries 8 --cpu-budget-s 10 --cpu-vec-queries 8 --sweep-launches 10 > "$OUT/bench_cfg3_plan_$PLAN.json" 2> "$OUT/bench_cfg3_plan_$PLAN.err"
  python - "$PLAN" <<'P'
import json,sys
d=json.load(open(f"gpurun_out/r05a/bench_cfg3_plan_{sys.argv[1]}.json"))
print("plan", sys.argv[1], d["value"], d["ms_per_step"], d["phases_ms"]["ppr_ms"], d["parity_spot_check"]["max_rel_score_err"], d["parity_spot_check"]["exact_id_fraction"], d["ppr_contract"]["ppr_residual_max"])
P
done
for LOC in none auto; do
  timeout 600 python bench.py --config real2wiki --locality $LOC --steps 20 --warmup 5 --cpu-queries 6 --cpu-budget-s 15 --cpu-vec-queries 8 > "$OUT/bench_real2wiki_$LOC.json" 2> "$OUT/bench_real2wiki_$LOC.err"
  python - "$LOC" <<'P'
import json,sys
d=json.load(open(f"gpurun_out/r05a/bench_real2wiki_{sys.argv[1]}.json"))
print("real2wiki", sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms_by_mode"], d["config"]["locality_score_after_renumbering"], d["parity_spot_check"]["max_rel_score_err"], d["parity_spot_check"]["exact_id_fraction"])
P
done
du -sh "$OUT"
