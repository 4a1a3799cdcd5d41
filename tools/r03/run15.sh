#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03r
timeout 900 python tools/bench_knn.py --n 875000 --nq 100000 --ref-queries 2000 > gpurun_out/r03r/bench_knn_final.json 2>/dev/null; cat gpurun_out/r03r/bench_knn_final.json | cut -c1-1500
timeout 300 python -m pytest tests/test_gpu_knn.py -q -m gpu 2>&1 | tail -2
