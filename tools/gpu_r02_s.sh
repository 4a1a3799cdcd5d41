#!/bin/bash
# round 2: pair kernel as the default -- whole GPU suite, cfg3 / world-1 / cfg4gpu / cfg5gpu lines
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02s}
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 --tb=short > "$OUT/gpu_tests.log" 2>&1
tail -6 "$OUT/gpu_tests.log"
for B in 256 384; do
timeout 600 python bench.py --config cfg3 --batch $B --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_cfg3_b$B.json" 2> "$OUT/bench_cfg3_b$B.err"
python - <<PY
import json
d=json.load(open("$OUT/bench_cfg3_b$B.json")); print("cfg3 B=$B", round(d["value"]), round(d["ms_per_step"],3), round(d["phases_ms"]["ppr_ms"],3), round(d["roofline"]["frac"],4), {k: round(v,4) for k,v in d["roofline"]["launch_ms_by_mode"].items()})
PY
done
MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 HRAG_FORCE_DIST=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench_dist_w1.json" 2> "$OUT/bench_dist_w1.err"
python - <<PY
import json
t=open("$OUT/bench_dist_w1.json").read().strip().splitlines(); d=json.loads([l for l in t if l.startswith("{")][-1])
print("dist w1 rowshard", d["rowshard"]["ms_per_step"], "replica", d["replica"]["ms_per_step"], d["rowshard"]["parity"]["ok"], d["rowshard"].get("exchange_groups"))
PY
timeout 600 python bench.py --config cfg4gpu --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench_cfg4gpu.json" 2> "$OUT/bench_cfg4gpu.err"
python - <<PY
import json
d=json.load(open("$OUT/bench_cfg4gpu.json")); print("cfg4gpu", round(d["value"]), round(d["ms_per_step"],3))
PY
