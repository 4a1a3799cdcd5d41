#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03r
timeout 600 python -m pytest tests/test_gpu_knn.py -q -m gpu > gpurun_out/r03r/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03r/tests.log; grep -n "^E  " gpurun_out/r03r/tests.log | head
timeout 600 python tools/bench_knn.py > gpurun_out/r03r/bench_knn_100k.json 2> gpurun_out/r03r/bench_knn_100k.err; cat gpurun_out/r03r/bench_knn_100k.json; tail -3 gpurun_out/r03r/bench_knn_100k.err
timeout 900 python tools/bench_knn.py --n 875000 --nq 20000 --ref-queries 2000 > gpurun_out/r03r/bench_knn_875k.json 2> gpurun_out/r03r/bench_knn_875k.err; cat gpurun_out/r03r/bench_knn_875k.json; tail -3 gpurun_out/r03r/bench_knn_875k.err
