#!/bin/bash
# measurement pass of a round: whole GPU suite, the bench line of every configuration (CPU baselines included), the
# N > 1 code path on one GPU (world 1), small-batch latencies, the randomised soaks, rocprofv3 kernel stats + PMC passes
# of the cfg 3 command
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04z}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 --tb=short > "$OUT/gpu_tests.log" 2>&1
tail -3 "$OUT/gpu_tests.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
tail -c 300 "$OUT/bench_cfg3.json"; echo
for CFG in cfg1s cfg2; do
  timeout 600 python bench.py --config $CFG --steps 50 --warmup 5 --cpu-queries 16 > "$OUT/bench_$CFG.json" 2> "$OUT/bench_$CFG.err"
  tail -c 200 "$OUT/bench_$CFG.json"; echo
done
HRAG_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_dist_world1.json" 2> "$OUT/bench_dist_world1.err"
HRAG_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_cfg4_world1.json" 2> "$OUT/bench_cfg4_world1.err"
timeout 600 python tools/sweep_smallb.py --batches 1,2,4,8,16,32 --out "$OUT/sweep_smallb.json" > "$OUT/sweep_smallb.log" 2> "$OUT/sweep_smallb.err"
cut -c1-80 "$OUT/sweep_smallb.log"
# randomised differential soaks (each prints one line per case and SOAK OK / SOAK FAILED)
for S in soak_random soak_shards soak_mirror soak_knn; do
  timeout 400 python tools/$S.py --seconds ${SOAK_SECONDS:-60} --seed ${SOAK_SEED:-1} > "$OUT/$S.log" 2>&1
  echo "$S: $(tail -1 "$OUT/$S.log")"
done
bash tools/gpu_profile.sh "$TAG/prof" --steps 5 --warmup 1 > "$OUT/profile.log" 2>&1
tail -3 "$OUT/profile.log"
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
du -sh "$OUT"
