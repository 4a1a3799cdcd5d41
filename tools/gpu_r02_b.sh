#!/bin/bash
# round 2, second GPU pass: new parity tests, secondary configs, rocprof stats + PMC of the new PPR kernels
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02b}
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_fp8_adversarial.py tests/test_mirror_surface.py -q -m gpu --maxfail=12 --tb=short > "$OUT/new_tests.log" 2>&1
tail -25 "$OUT/new_tests.log"
for CFG in cfg1s cfg2; do
  timeout 400 python bench.py --config $CFG --steps 50 --warmup 5 --cpu-queries 16 > "$OUT/bench_$CFG.json" 2> "$OUT/bench_$CFG.err"
  tail -c 300 "$OUT/bench_$CFG.json"; tail -2 "$OUT/bench_$CFG.err"
done
timeout 400 python bench.py --config cfg4gpu --steps 4 --warmup 1 > "$OUT/bench_cfg4gpu.json" 2> "$OUT/bench_cfg4gpu.err"
tail -c 900 "$OUT/bench_cfg4gpu.json"; tail -2 "$OUT/bench_cfg4gpu.err"
bash tools/gpu_profile.sh "$(basename $OUT)/prof" --steps 5 --warmup 1 > "$OUT/profile.log" 2>&1
tail -5 "$OUT/profile.log"
