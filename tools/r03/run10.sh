#!/bin/bash
# locality experiment: the sweep on a graph with community structure, SELL-C-sigma + blocked XCD mapping; bench + TCC hit rate
set -x
cd "$GRAFT_REPO_ROOT"
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03n
for V in "0 0" "4096 2048" "16384 2048" "4096 0"; do
  set -- $V
  timeout 600 python bench.py --config cfg3loc --steps 10 --warmup 2 --no-cpu-baseline --sell-sigma $1 --engine-flags $2 > gpurun_out/r03n/bench_loc_s$1_f$2.json 2> gpurun_out/r03n/bench_loc_s$1_f$2.err; echo "rc=$?"
done
timeout 600 python bench.py --config cfg3 --steps 10 --warmup 2 --no-cpu-baseline --sell-sigma 4096 --engine-flags 2048 > gpurun_out/r03n/bench_cfg3_s4096_f2048.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03n/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d["value"]), round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],4), "C", round(d["roofline"]["launch_ms_by_mode"]["C"],4), "ppr", round(d["phases_ms"]["ppr_ms"],3))
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
for S in "0 0" "4096 2048"; do
  set -- $S
  HRAG_PMC_CONFIG=cfg3loc HRAG_SELL_SIGMA=$1 HRAG_FLAGS=$2 timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $REPO/gpurun_out/r03n/pmc_hit_s$1 -o pmc -- python $REPO/tools/pmc_target.py > $REPO/gpurun_out/r03n/pmc_hit_s$1.log 2>&1
  HRAG_PMC_CONFIG=cfg3loc HRAG_SELL_SIGMA=$1 HRAG_FLAGS=$2 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/r03n/pmc_fetch_s$1 -o pmc -- python $REPO/tools/pmc_target.py > $REPO/gpurun_out/r03n/pmc_fetch_s$1.log 2>&1
done
cd $REPO
python - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob("gpurun_out/r03n/pmc_*_s*/")):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(d+"**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "ppr8_pair_kernel" in r["Kernel_Name"]:
                k=r["Kernel_Name"].split("ppr8_pair_kernel")[1][:12]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(d, k, {c: sum(x)/len(x) for c,x in v.items()})
PY
find gpurun_out/r03n -name "*kernel_trace.csv" -delete
