#!/bin/bash
# round 2: pair kernel vs one slab per wavefront on the hub-heavy power-law share (cfg5gpu), same (pair) layout
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02v}
mkdir -p "$OUT"
cd "$REPO"
for P in 0 1; do
  HRAG_P8_PAIR=$P timeout 900 python bench.py --config cfg5gpu --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_cfg5gpu_pair$P.json" 2> "$OUT/bench_cfg5gpu_pair$P.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_cfg5gpu_pair$P.json")); print("cfg5gpu pair $P", round(d["value"]), round(d["phases_ms"]["ppr_ms"],1), {k: round(v,2) for k,v in d["roofline"]["launch_ms_by_mode"].items()})
PY
done
