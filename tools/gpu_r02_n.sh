#!/bin/bash
# round 2: fp16-state sweep with the row inputs loaded before the gather loop
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02n}
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long_rows.py tests/test_ref_golden.py -q -m gpu --maxfail=8 --tb=short > "$OUT/gpu_tests.log" 2>&1
tail -4 "$OUT/gpu_tests.log"
for C in cfg1s cfg2; do
  timeout 600 python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$C.json")); print("$C", round(d["value"]), round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, round(d["roofline"]["frac"],4))
PY
done
timeout 600 python tools/sweep_smallb.py --batches 16,64 --out "$OUT/sweep_smallb.json" > "$OUT/sweep_smallb.log" 2> "$OUT/sweep_smallb.err"
cut -c1-60 "$OUT/sweep_smallb.log"
