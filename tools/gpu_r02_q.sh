#!/bin/bash
# round 2: state layout of the single-GPU fp8 path: slab-major (128-byte gathers) vs vertex-major (the slabs of a
# vertex adjacent: the wavefronts of a workgroup fetch one 256 / 512-byte piece together)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02q}
mkdir -p "$OUT"
cd "$REPO"
for G in 0 1; do
 for B in 256 512; do
  HRAG_P8_GROUPS=$G timeout 600 python bench.py --config cfg3 --batch $B --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/bench_g${G}_b$B.json" 2> "$OUT/bench_g${G}_b$B.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_g${G}_b$B.json")); print("groups $G batch $B", round(d["value"]), round(d["ms_per_step"],3), round(d["phases_ms"]["ppr_ms"],3), d["roofline"]["launch_ms_by_mode"])
PY
 done
done
