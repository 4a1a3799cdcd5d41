#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_shard.py tests/test_gpu_fp8_adversarial.py -q -m gpu > gpurun_out/r03o_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r03o_tests.log
grep -n "Error\|^E  " gpurun_out/r03o_tests.log | head -30
