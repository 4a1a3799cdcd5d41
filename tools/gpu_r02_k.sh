#!/bin/bash
# round 2: SELL-8 segment length (longest row = the critical path of a sweep) -- cfg2, cfg3, cfg3 B=1
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02k}
mkdir -p "$OUT"
cd "$REPO"
for L in 64 128 256 512; do
  HRAG_SELL8_SEG_LEN=$L timeout 600 python bench.py --config cfg2 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_cfg2_$L.json" 2> "$OUT/bench_cfg2_$L.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_cfg2_$L.json")); print("cfg2 seg $L", round(d["value"]), round(d["ms_per_step"],3), round(d["phases_ms"]["ppr_ms"],3), round(d["roofline"]["frac"],4))
PY
done
for L in 128 256 512 1024; do
  HRAG_SELL8_SEG_LEN=$L timeout 600 python bench.py --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/bench_cfg3_$L.json" 2> "$OUT/bench_cfg3_$L.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_cfg3_$L.json")); print("cfg3 seg $L", round(d["value"]), round(d["ms_per_step"],3), round(d["phases_ms"]["ppr_ms"],3), round(d["roofline"]["frac"],4))
PY
  HRAG_SELL8_SEG_LEN=$L timeout 600 python tools/sweep_smallb.py --batches 1,8,32 --out "$OUT/sweep_smallb_$L.json" > "$OUT/sweep_smallb_$L.log" 2>&1
  cut -c1-60 "$OUT/sweep_smallb_$L.log" | grep latency
done
