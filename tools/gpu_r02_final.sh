#!/bin/bash
# round 2, final measurement pass: whole GPU suite, the bench line of every configuration (CPU baselines included),
# row-shard world-1 line, small-batch latencies, rocprofv3 kernel stats + PMC passes of the cfg 3 command, GEMM PMC
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02z}
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 --tb=short > "$OUT/gpu_tests.log" 2>&1
tail -5 "$OUT/gpu_tests.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
tail -c 300 "$OUT/bench_cfg3.json"; echo
for CFG in cfg1s cfg2; do
  timeout 600 python bench.py --config $CFG --steps 50 --warmup 5 --cpu-queries 16 > "$OUT/bench_$CFG.json" 2> "$OUT/bench_$CFG.err"
  tail -c 200 "$OUT/bench_$CFG.json"; echo
done
MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 HRAG_FORCE_DIST=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench_dist_w1.json" 2> "$OUT/bench_dist_w1.err"
tail -c 300 "$OUT/bench_dist_w1.json"; echo
timeout 600 python bench.py --config cfg4gpu --steps 4 --warmup 1 > "$OUT/bench_cfg4gpu.json" 2> "$OUT/bench_cfg4gpu.err"
tail -c 300 "$OUT/bench_cfg4gpu.json"; echo
timeout 600 python tools/sweep_smallb.py --batches 1,2,4,8,16,32 --out "$OUT/sweep_smallb.json" > "$OUT/sweep_smallb.log" 2> "$OUT/sweep_smallb.err"
cut -c1-60 "$OUT/sweep_smallb.log"
bash tools/gpu_profile.sh "$(basename $OUT)/prof" --steps 5 --warmup 1 > "$OUT/profile.log" 2>&1
tail -3 "$OUT/profile.log"
bash tools/gpu_profile_gemm.sh "$(basename $OUT)/gemm" > "$OUT/gemm_profile.log" 2>&1
tail -12 "$OUT/gemm_profile.log"
timeout 1500 python bench.py --config cfg5gpu --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_cfg5gpu.json" 2> "$OUT/bench_cfg5gpu.err"
tail -c 300 "$OUT/bench_cfg5gpu.json"; echo
# large raw traces stay on the box: keep the summaries
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
du -sh "$OUT"
