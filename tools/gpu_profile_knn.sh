#!/bin/bash
# Kernel-trace stats + PMC passes for the KNN similarity GEMM at its MFMA-bound shape: bash tools/gpu_profile_knn.sh <tag>
set -u
TAG=${1:-knn}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o knn -- \
    python "$REPO/tools/pmc_knn_target.py" > "$OUT/trace.log" 2>&1
tail -2 "$OUT/trace.log"
for C in "MfmaUtil" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16"; do
    N=$(echo $C | tr ' ' '_')
    timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$N" -o pmc -- \
        python "$REPO/tools/pmc_knn_target.py" > "$OUT/pmc_$N.log" 2>&1
    tail -1 "$OUT/pmc_$N.log"
done
cd "$REPO"
python - <<PY
import csv, glob, collections, json
out = {}
for path in sorted(glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(path, newline="")):
        if any(s in r["Name"] for s in ("sim_gemm", "tile_", "split3", "row_topk")):
            out.setdefault("kernel_stats", {})[r["Name"][:60]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                                   "total_ns": float(r["TotalDurationNs"])}
for path in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path, newline="")):
        if "sim_gemm" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out.setdefault("pmc", {}).setdefault(k, {})[c] = sum(v) / len(v)
n, nq, d = int("${HRAG_KNN_KEYS:-875000}"), 4096, 2304
for k, v in out.get("kernel_stats", {}).items():
    if "sim_gemm256_kernel<8, true, true>" in k:
        v["tflops_at_4096_queries_per_launch"] = 2.0 * n * nq * d / v["avg_ns"] / 1e3
json.dump(out, open("$OUT/knn_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
