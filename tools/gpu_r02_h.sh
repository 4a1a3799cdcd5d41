#!/bin/bash
# round 2: the B <= 8 path (fp16 two-stage state, masked first sweep, passage-only last sweep) -- parity + latency
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02h}
mkdir -p "$OUT"
cd "$REPO"
timeout 1200 python -m pytest tests -q -m gpu --maxfail=8 --tb=short > "$OUT/gpu_tests.log" 2>&1
tail -12 "$OUT/gpu_tests.log"
timeout 600 python tools/sweep_smallb.py --batches 1,2,4,8,16,32 --out "$OUT/sweep_smallb.json" > "$OUT/sweep_smallb.log" 2> "$OUT/sweep_smallb.err"
tail -8 "$OUT/sweep_smallb.log"; tail -3 "$OUT/sweep_smallb.err"
