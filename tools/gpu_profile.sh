#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of bench.py + separate PMC passes for the SpMM kernel.
#   bash tools/gpu_profile.sh <tag> [bench args...]
# Outputs under gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ -z "${HRAG_PMC_ONLY:-}" ]; then
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
    python "$REPO/bench.py" --no-cpu-baseline --ppr-tol 0 "$@" > "$OUT/bench_under_rocprof.log" 2>&1
fi
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    N=$(echo $C | tr ' ' '_')
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$N" -o pmc -- \
        python "$REPO/tools/pmc_target.py" > "$OUT/pmc_$N.log" 2>&1
done
cd "$REPO"
find "$OUT" -name '*.csv' | head -50
