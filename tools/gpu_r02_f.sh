#!/bin/bash
# round 2: the 256-row GEMM tile -- parity tests + phase timings with / without it + MFMA utilisation PMC
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02f}
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_knn.py tests/test_ref_golden.py -q -m gpu --maxfail=8 --tb=short -k "sim or fused or score or knn or golden or end_to_end or dense" > "$OUT/gemm_tests.log" 2>&1
tail -15 "$OUT/gemm_tests.log"
for SMALL in 1 0; do
  HRAG_SIM_SMALL_TILES=$SMALL timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/bench_small$SMALL.json" 2> "$OUT/bench_small$SMALL.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_small$SMALL.json")); print("small_tiles=$SMALL", round(d["value"]), d["phases_ms"])
PY
done
bash tools/gpu_profile_gemm.sh "$(basename $OUT)/gemm" > "$OUT/gemm_profile.log" 2>&1
tail -5 "$OUT/gemm_profile.log"
