#!/bin/bash
# round 2: fused fact top-k, pass 3 as one workgroup per (tile, query)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02o}
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 --tb=short > "$OUT/gpu_tests.log" 2>&1
tail -6 "$OUT/gpu_tests.log"
for C in cfg1s cfg2 cfg3; do
  timeout 600 python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$C.json")); print("$C", round(d["value"]), round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, round(d["roofline"]["frac"],4))
PY
done
