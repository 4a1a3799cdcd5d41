#!/bin/bash
# round 3, call 3: bench lines with the contract legs
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r03c_bench_cfg3.json 2> gpurun_out/r03c_bench_cfg3.err; echo "cfg3 rc=$?"
timeout 600 python bench.py --config cfg2 --steps 50 --warmup 5 > gpurun_out/r03c_bench_cfg2.json 2> gpurun_out/r03c_bench_cfg2.err; echo "cfg2 rc=$?"
tail -3 gpurun_out/r03c_bench_cfg3.err
python - <<'PY'
import json
for c in ("cfg3","cfg2"):
    try:
        d=json.loads(open(f"gpurun_out/r03c_bench_{c}.json").read().strip().splitlines()[-1])
        print(c, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["ppr_contract"], d["with_convergence_contract"], d.get("parity_spot_check"), d["phases_ms"])
    except Exception as e:
        print(c, "ERR", e)
PY
