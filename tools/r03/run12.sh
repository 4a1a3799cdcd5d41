#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03p
timeout 900 python -m pytest tests/test_gpu_locality.py tests/test_mirror_surface.py tests/test_golden.py tests/test_incremental_index.py -q -m gpu > gpurun_out/r03p/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03p/tests.log; grep -n "^E  " gpurun_out/r03p/tests.log | head -20
timeout 600 python bench.py --config cfg3hash --steps 10 --warmup 2 --no-cpu-baseline --ppr-tol 0 > gpurun_out/r03p/bench_hash_none.json 2> gpurun_out/r03p/bench_hash_none.err
timeout 600 python bench.py --config cfg3hash --steps 10 --warmup 2 --cpu-queries 6 --ppr-tol 0 --locality auto > gpurun_out/r03p/bench_hash_auto.json 2> gpurun_out/r03p/bench_hash_auto.err
timeout 600 python bench.py --config cfg3 --steps 10 --warmup 2 --no-cpu-baseline --ppr-tol 0 --locality auto > gpurun_out/r03p/bench_cfg3_auto.json 2> gpurun_out/r03p/bench_cfg3_auto.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03p/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d["value"]), round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],4), "C", round(d["roofline"]["launch_ms_by_mode"]["C"],4), "score", d["config"]["locality_score_after_renumbering"], "flags", d["config"]["engine_opt_flags"], "setup", round(d["setup_s"],1), d.get("parity_spot_check"))
    except Exception as e: print(f, "ERR", e)
PY
