#!/bin/bash
# round 3: re-pin parity at the big sizes on the round-3 kernels
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_shard.py -x -q -m gpu > gpurun_out/r03i_shard_tests.log 2>&1; echo "shard tests rc=$?"; tail -3 gpurun_out/r03i_shard_tests.log
(time timeout 900 python bench.py --config cfg4local --cpu-queries 16) > gpurun_out/r03i_bench_cfg4local.json 2> gpurun_out/r03i_bench_cfg4local.err; echo "cfg4local rc=$?"
tail -c 1500 gpurun_out/r03i_bench_cfg4local.json; tail -5 gpurun_out/r03i_bench_cfg4local.err
free -g | head -2
(time timeout 1500 python bench.py --config cfg5gpu --steps 3 --warmup 1 --cpu-queries 8 --cpu-budget-s 400 --sweep-launches 5) > gpurun_out/r03i_bench_cfg5.json 2> gpurun_out/r03i_bench_cfg5.err; echo "cfg5 rc=$?"
tail -c 3000 gpurun_out/r03i_bench_cfg5.json; tail -5 gpurun_out/r03i_bench_cfg5.err
