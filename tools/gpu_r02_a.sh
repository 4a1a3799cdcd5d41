#!/bin/bash
# round 2, first GPU pass: shard tests, the whole GPU suite, cfg3 bench, world-1 dist bench
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02a}
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_shard.py -q -m gpu --maxfail=6 --tb=short > "$OUT/shard_tests.log" 2>&1
tail -5 "$OUT/shard_tests.log"
timeout 1200 python -m pytest tests -q -m gpu --maxfail=10 --tb=short --deselect tests/test_gpu_shard.py > "$OUT/gpu_tests.log" 2>&1
tail -5 "$OUT/gpu_tests.log"
timeout 600 python bench.py --steps 10 --warmup 2 > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
tail -c 600 "$OUT/bench_cfg3.json"; tail -3 "$OUT/bench_cfg3.err"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 HRAG_FORCE_DIST=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench_dist_w1.json" 2> "$OUT/bench_dist_w1.err"
tail -c 1500 "$OUT/bench_dist_w1.json"; tail -3 "$OUT/bench_dist_w1.err"
