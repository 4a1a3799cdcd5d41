#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
HRAG_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03m_bench_dist_world1.json 2> gpurun_out/r03m_bench_dist_world1.err; echo "dist rc=$?"
tail -c 2500 gpurun_out/r03m_bench_dist_world1.json; tail -3 gpurun_out/r03m_bench_dist_world1.err
(time timeout 900 python bench.py --config cfg4local --cpu-queries 16) > gpurun_out/r03m_bench_cfg4local.json 2> gpurun_out/r03m_bench_cfg4local.err; echo "cfg4local rc=$?"
tail -c 1200 gpurun_out/r03m_bench_cfg4local.json
