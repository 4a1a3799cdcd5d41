#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fp8_adversarial.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03h_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r03h_tests.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r03h_bench_cfg3.json 2> gpurun_out/r03h_bench_cfg3.err; echo "cfg3 rc=$?"
timeout 600 python bench.py --config cfg2 --steps 50 --warmup 5 > gpurun_out/r03h_bench_cfg2.json 2> gpurun_out/r03h_bench_cfg2.err; echo "cfg2 rc=$?"
python - <<'PY'
import json
for c in ("cfg3","cfg2"):
    try:
        d=json.loads(open(f"gpurun_out/r03h_bench_{c}.json").read().strip().splitlines()[-1])
        pc, wc = d["ppr_contract"], d["with_convergence_contract"]
        print(c, round(d["value"]), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), "resid", pc["ppr_residual_max"], "| contract:", round(wc["value"]), wc["sweeps_used_max"], wc["ppr_residual_max"], wc["queries_flagged_not_converged"], d["parity_spot_check"]["max_rel_score_err"], d["phases_ms"], d["roofline"]["launch_ms_by_mode"])
    except Exception as e:
        print(c, "ERR", e)
PY
