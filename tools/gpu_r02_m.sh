#!/bin/bash
# round 2: fp16-state path (8 < B <= 64) with masked first sweep / passage-only last sweep / closed-form mass
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02m}
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 --tb=short > "$OUT/gpu_tests.log" 2>&1
tail -8 "$OUT/gpu_tests.log"
for C in cfg1s cfg2; do
  timeout 600 python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$C.json")); print("$C", round(d["value"]), round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, round(d["roofline"]["frac"],4))
PY
done
timeout 600 python tools/sweep_smallb.py --batches 16,32,64 --out "$OUT/sweep_smallb.json" > "$OUT/sweep_smallb.log" 2> "$OUT/sweep_smallb.err"
cut -c1-60 "$OUT/sweep_smallb.log"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t_cfg2" -o t -- python "$REPO/tools/trace_target.py" --config cfg2 > "$OUT/t_cfg2.log" 2>&1
cd "$REPO"
python tools/timeline.py "$(find "$OUT/t_cfg2" -name '*kernel_trace.csv' | head -1)" > "$OUT/t_cfg2.timeline.txt" 2>&1
find "$OUT/t_cfg2" -name '*.csv' -size +2M -delete
tail -2 "$OUT/t_cfg2.timeline.txt"
