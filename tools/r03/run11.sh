#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03n
for V in "0 0" "4096 2048" "16384 2048" "1024 2048" "4096 0"; do
  set -- $V
  timeout 600 python bench.py --config cfg3loc --steps 10 --warmup 2 --no-cpu-baseline --sell-sigma $1 --engine-flags $2 > gpurun_out/r03n/bench_loc_s$1_f$2.json 2> gpurun_out/r03n/bench_loc_s$1_f$2.err; echo "rc=$?"
done
timeout 600 python bench.py --config cfg3 --steps 10 --warmup 2 --no-cpu-baseline --sell-sigma 4096 --engine-flags 2048 > gpurun_out/r03n/bench_cfg3_s4096_f2048.json 2>/dev/null
timeout 600 python bench.py --config cfg3loc --steps 10 --warmup 2 --cpu-queries 8 --sell-sigma 4096 --engine-flags 2048 > gpurun_out/r03n/bench_loc_s4096_f2048_oracle.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03n/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d["value"]), round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],4), "C", round(d["roofline"]["launch_ms_by_mode"]["C"],4), "ppr", round(d["phases_ms"]["ppr_ms"],3), d.get("parity_spot_check"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python -m pytest tests/test_mirror_surface.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -3
