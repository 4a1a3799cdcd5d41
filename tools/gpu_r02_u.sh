#!/bin/bash
# round 2: pair kernel, (col, val) stream non-temporal when a single pair reads it
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02u}
mkdir -p "$OUT"
cd "$REPO"
for NT in 0 1 0 1; do
  HRAG_P8_PAIR_NT=$NT timeout 600 python bench.py --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/bench_nt$NT.json" 2> "$OUT/bench_nt$NT.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_nt$NT.json")); print("nt $NT", round(d["value"]), round(d["phases_ms"]["ppr_ms"],3), {k: round(v,4) for k,v in d["roofline"]["launch_ms_by_mode"].items()})
PY
done
