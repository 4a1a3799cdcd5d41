#!/bin/bash
# kernel-trace stats of the cfg3 bench (both legs: fixed 20 sweeps, convergence contract)
set -x
cd "$GRAFT_REPO_ROOT"
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r03g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python "$REPO/bench.py" --no-cpu-baseline --steps 20 --warmup 3 > "$OUT/bench_under_rocprof.log" 2>&1
echo rc=$?
cd $REPO
python tools/prof_summary.py gpurun_out/r03g gpurun_out/r03g_cfg3 || true
head -40 gpurun_out/r03g_cfg3_kernel_stats.csv
rm -rf $OUT/trace/*/*kernel_trace.csv
