#!/bin/bash
# PMC passes for the similarity GEMM (MFMA utilisation, HBM fetch): bash tools/gpu_profile_gemm.sh <tag>
set -u
TAG=${1:-gemm}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in "MfmaUtil" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "FETCH_SIZE"; do
    N=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$N" -o pmc -- \
        python "$REPO/tools/pmc_gemm_target.py" > "$OUT/pmc_$N.log" 2>&1
    tail -2 "$OUT/pmc_$N.log"
done
cd "$REPO"
python - <<PY
import csv, glob, collections
for path in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path, newline="")):
        if "sim_gemm" in r["Kernel_Name"] or "tile_" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, sum(v) / len(v), len(v))
PY
