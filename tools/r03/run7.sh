#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ref_golden.py tests/test_loaders.py tests/test_incremental_index.py tests/test_golden.py tests/test_mirror_surface.py -q -m gpu > gpurun_out/r03k_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r03k_tests.log
grep -n "^E  .*assert\|Max absolute\|Max relative\|AssertionError: " gpurun_out/r03k_tests.log | head -30
