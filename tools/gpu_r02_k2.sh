#!/bin/bash
# round 2: SELL-8 segment length, second pass: shorter segments on the small graphs, wide batches on the small graph,
# cfg4gpu / cfg5gpu shares
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02k2}
mkdir -p "$OUT"
cd "$REPO"
run() {  # config seglen extra-args...
  local C=$1 L=$2; shift 2
  local TAG="${C}_${L}$(echo "$*" | tr -d ' -')"
  HRAG_SELL8_SEG_LEN=$L timeout 900 python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$TAG.json")); print("$TAG", round(d["value"]), round(d["ms_per_step"],3), round(d["phases_ms"]["ppr_ms"],3), round(d["roofline"]["frac"],4))
PY
}
for L in 16 32 64; do run cfg2 $L; done
for L in 16 32 64 512; do run cfg1s $L; done
for L in 32 64 128 512; do run cfg2 $L --batch 256; done
for L in 32 64 128; do run cfg2 $L --batch 16; done
for L in 256 512; do run cfg4gpu $L; done
