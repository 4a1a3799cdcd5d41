#!/bin/bash
# round 2, third GPU pass: whole GPU suite on the new kernels (masked B0, 3-byte residual), cfg3 bench + dist world 1
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02c}
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 --tb=short > "$OUT/gpu_tests.log" 2>&1
tail -30 "$OUT/gpu_tests.log"
timeout 600 python bench.py --steps 10 --warmup 2 > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
tail -c 400 "$OUT/bench_cfg3.json"; tail -3 "$OUT/bench_cfg3.err"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 HRAG_FORCE_DIST=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench_dist_w1.json" 2> "$OUT/bench_dist_w1.err"
tail -c 700 "$OUT/bench_dist_w1.json"; tail -3 "$OUT/bench_dist_w1.err"
