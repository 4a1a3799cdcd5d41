#!/bin/bash
# round 3, call 1: smoke + the fp8 tests on the new plan code + the convergence probe
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r03a_smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests/test_gpu_fp8_adversarial.py tests/test_gpu_shard.py -x -q -m gpu > gpurun_out/r03a_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r03a_tests.log
timeout 600 python tools/r03/conv_probe.py > gpurun_out/r03a_probe.log 2>&1; echo "probe rc=$?"
cat gpurun_out/r03a_probe.log | tail -30
