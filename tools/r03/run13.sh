#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03s
timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 --tb=short > gpurun_out/r03s/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03s/tests.log; grep -n "^E  \|^FAILED" gpurun_out/r03s/tests.log | head -20
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03s/bench_cfg3.json 2>/dev/null
timeout 600 python bench.py --config cfg2 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r03s/bench_cfg2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03s/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d["value"]), round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],4), d["phases_ms"], d["with_convergence_contract"].get("value"))
PY
