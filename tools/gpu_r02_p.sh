#!/bin/bash
# round 2: segment length on the hub-heavy power-law share (cfg5gpu)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02p}
mkdir -p "$OUT"
cd "$REPO"
for L in 2048 8192; do
  HRAG_SELL8_SEG_LEN=$L timeout 900 python bench.py --config cfg5gpu --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_cfg5gpu_$L.json" 2> "$OUT/bench_cfg5gpu_$L.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_cfg5gpu_$L.json")); print("cfg5gpu seg $L", round(d["value"]), round(d["phases_ms"]["ppr_ms"],1), d["roofline"]["launch_ms_by_mode"])
PY
done
