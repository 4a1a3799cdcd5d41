#!/bin/bash
# round 2: pair kernel (two adjacent slabs per wavefront, vertex-major state) vs one slab per wavefront
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02r}
mkdir -p "$OUT"
cd "$REPO"
HRAG_P8_GROUPS=1 HRAG_P8_PAIR=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fp8_adversarial.py tests/test_gpu_long_rows.py -q -m gpu --tb=short -k "f8 or fp8 or long or full_size" > "$OUT/pair_tests.log" 2>&1
tail -4 "$OUT/pair_tests.log"
for P in 0 1; do
 for B in 256 512; do
  HRAG_P8_GROUPS=1 HRAG_P8_PAIR=$P timeout 600 python bench.py --config cfg3 --batch $B --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/bench_p${P}_b$B.json" 2> "$OUT/bench_p${P}_b$B.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_p${P}_b$B.json")); print("pair $P batch $B", round(d["value"]), round(d["ms_per_step"],3), round(d["phases_ms"]["ppr_ms"],3), {k: round(v,4) for k,v in d["roofline"]["launch_ms_by_mode"].items()})
PY
 done
done
