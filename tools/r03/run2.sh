#!/bin/bash
# round 3, call 2: the whole GPU suite on the convergence-contract build
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r03j_tests.log 2>&1; echo "tests rc=$?"
tail -30 gpurun_out/r03j_tests.log
