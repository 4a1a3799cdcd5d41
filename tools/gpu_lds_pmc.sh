#!/bin/bash
# LDS / wait counters of the similarity GEMM kernels (tools/pmc_gemm_target.py)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-lds}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "LDS|WAIT_INST|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_INST_CYCLES_VMEM|SQ_WAIT_ANY|MfmaUtil|SQ_ACTIVE_INST" | head -60 > "$OUT/avail.txt"
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"; do
    N=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$N" -o pmc -- \
        python "$REPO/tools/pmc_gemm_target.py" > "$OUT/pmc_$N.log" 2>&1
done
cd "$REPO"
python - <<PY
import csv, glob, collections
for path in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path, newline="")):
        if "sim_gemm" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][40:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, sum(v) / len(v), len(v))
PY
