#!/bin/bash
# measurement pass of a round: whole GPU suite + smoke, the bench line of every configuration (CPU baselines included),
# the real-topology graph with and without the locality numbering, the N > 1 code path on one GPU (world 1: oracle probe,
# configs[3] strong legs, both exchange collectives), small-batch latencies, the mirror end to end, the randomised soaks,
# rocprofv3 kernel stats + PMC passes of the cfg 3 command.      bash tools/measure_round.sh <tag>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06z}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
( time timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 --tb=short --durations=8 ) > "$OUT/gpu_tests.log" 2>&1
tail -16 "$OUT/gpu_tests.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
python - "$TAG" <<'P'
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}/bench_cfg3.json")); r=d["roofline"]
print("cfg3:", round(d["value"]), d["ms_per_step"], "frac", r["frac"], "harness", r["launch_ms_from_instantiation_harness"], r["instantiation_harness_agrees"], "traffic", r["traffic"],
      "err", d["parity_spot_check"]["max_rel_score_err"], "resid", d["ppr_contract"]["ppr_residual_max"], d["ppr_contract"]["meets_default_tol"],
      "contract", d["with_convergence_contract"].get("value"), d["with_convergence_contract"].get("sweeps_used_max"),
      "accel", d["with_accelerated_stages"].get("value"), d["with_accelerated_stages"].get("with_convergence_contract",{}).get("value"))
P
for CFG in cfg1s cfg2; do
  timeout 600 python bench.py --config $CFG --steps 50 --warmup 5 --cpu-queries 16 > "$OUT/bench_$CFG.json" 2> "$OUT/bench_$CFG.err"
  python - "$TAG" "$CFG" <<'P'
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}/bench_{sys.argv[2]}.json"))
print(sys.argv[2], round(d["value"]), d["roofline"]["frac"], d["parity_spot_check"]["max_rel_score_err"], d["with_accelerated_stages"].get("value"))
P
done
for LOC in none auto; do
  timeout 600 python bench.py --config real2wiki --locality $LOC --steps 20 --warmup 5 --cpu-queries 6 --cpu-budget-s 15 --cpu-vec-queries 8 > "$OUT/bench_real2wiki_$LOC.json" 2> "$OUT/bench_real2wiki_$LOC.err"
  python - "$TAG" "$LOC" <<'P'
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}/bench_real2wiki_{sys.argv[2]}.json")); r=d["roofline"]
print("real2wiki", sys.argv[2], round(d["value"]), d["ms_per_step"], "frac", r["frac"], "harness agrees", r["instantiation_harness_agrees"], "C", r["launch_ms_by_mode"]["C"], d["config"]["locality_score_after_renumbering"], d["config"]["engine_opt_flags"], d["parity_spot_check"]["max_rel_score_err"], d["parity_spot_check"]["exact_id_fraction"])
P
done
HRAG_FORCE_DIST=1 HRAG_STRONG_GLOBAL_BATCH=128 timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > "$OUT/bench_dist_world1.json" 2> "$OUT/bench_dist_world1.err"
python - "$TAG" <<'P'
import json,sys
try:
    d=json.load(open(f"gpurun_out/{sys.argv[1]}/bench_dist_world1.json"))
    print("world1:", round(d["value"]), d["value_leg"], d["value_rowshard"], d["value_hybrid"], d["value_replica"], d["rowshard"]["parity"]["ok"], d["rowshard"]["parity"].get("vs_oracle",{}).get("max_rel_score_err"), d["hybrid"]["parity"]["ok"], d.get("configs3_strong",{}).get("value_leg"))
except Exception as e:
    print("world1 FAILED", e); print(open(f"gpurun_out/{sys.argv[1]}/bench_dist_world1.err").read()[-1500:])
P
HRAG_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --collective allreduce --no-cpu-baseline --no-strong > "$OUT/bench_dist_world1_allreduce.json" 2> "$OUT/bench_dist_world1_allreduce.err"
HRAG_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_cfg4_world1.json" 2> "$OUT/bench_cfg4_world1.err"
timeout 600 python tools/sweep_smallb.py --batches 1,2,4,8,16,32,64 --out "$OUT/sweep_smallb.json" > "$OUT/sweep_smallb.log" 2> "$OUT/sweep_smallb.err"
cut -c1-100 "$OUT/sweep_smallb.log"
for CFG in cfg2 cfg3; do
  timeout 600 python tools/bench_mirror.py --config $CFG --queries 4096 > "$OUT/bench_mirror_$CFG.json" 2> "$OUT/bench_mirror_$CFG.err"; tail -c 300 "$OUT/bench_mirror_$CFG.json"; echo
done
# randomised differential soaks (each prints one line per case and SOAK OK / SOAK FAILED)
for S in soak_random soak_shards soak_mirror soak_knn; do
  timeout 400 python tools/$S.py --seconds ${SOAK_SECONDS:-90} --seed ${SOAK_SEED:-51} > "$OUT/$S.log" 2>&1
  echo "$S: $(tail -1 "$OUT/$S.log")"
done
bash tools/gpu_profile.sh "$TAG/prof" --steps 5 --warmup 1 > "$OUT/profile.log" 2>&1
tail -2 "$OUT/profile.log"
python tools/prof_summary.py "$OUT/prof" "$OUT/${TAG}_cfg3" > "$OUT/prof_summary.log" 2>&1; cat "$OUT/prof_summary.log"
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
du -sh "$OUT"
