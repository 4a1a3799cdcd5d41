#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 --tb=short > gpurun_out/r03t_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03t_tests.log; grep -n "^E  \|^FAILED" gpurun_out/r03t_tests.log | head -20
