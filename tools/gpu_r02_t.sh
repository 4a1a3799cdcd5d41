#!/bin/bash
# round 2: seeds kernel with batched loads, fewer top-k parts -- parity + small-batch latency + cfg2
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02t}
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 --tb=short > "$OUT/gpu_tests.log" 2>&1
tail -4 "$OUT/gpu_tests.log"
timeout 600 python tools/sweep_smallb.py --batches 1,2,8 --out "$OUT/sweep_smallb.json" > "$OUT/sweep_smallb.log" 2> "$OUT/sweep_smallb.err"
cut -c1-400 "$OUT/sweep_smallb.log"
for C in cfg2 cfg3; do
timeout 600 python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"
python - <<PY
import json
d=json.load(open("$OUT/bench_$C.json")); print("$C", round(d["value"]), round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()})
PY
done
