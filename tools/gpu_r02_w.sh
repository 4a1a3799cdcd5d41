#!/bin/bash
# round 2: PMC look at the cfg 2 sweep (ppr16_kernel): L2 hit rate, wave cycles vs waiting, fetched bytes
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02w}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_WAIT_ANY"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$N" -o pmc -- python "$REPO/tools/trace_target.py" --config cfg2 --calls 3 > "$OUT/pmc_$N.log" 2>&1
done
cd "$REPO"
python - <<PY
import csv, glob, collections
for path in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path, newline="")):
        if "ppr16_kernel" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("ppr16_kernel")[1][:22]
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, round(sum(v) / len(v), 1), len(v))
PY
find "$OUT" -name '*.csv' -size +1M -delete
