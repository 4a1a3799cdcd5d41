#!/usr/bin/env python
"""Condense the rocprofv3 CSV output of tools/gpu_profile.sh into small files for profiles/.

    python tools/prof_summary.py gpurun_out/<tag> profiles/<name>

writes  <name>_kernel_stats.csv   our kernels + top-5 others (names shortened), from --kernel-trace --stats
        <name>_pmc.json           per-kernel means of every PMC counter collected (separate passes) and
                                  the HBM traffic estimate for the SpMM kernels, corrected as
                                  /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
                                  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts the
                                  128-byte requests of wide (16 B / lane) reads at 64 B -> x2.
                                  WRITE_SIZE is uncalibrated (taken as reported).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^()]*?>)?)\(", name)
    out = m.group(1) if m else name
    return out[:120]


KERNEL_SOURCES = {"ppr8_kernel": "ppr8.hip", "ppr16_kernel": "ppr16.hip", "ppr_spmm_kernel": "ppr_spmm.hip"}


def kernel_source_sha16(kernel: str):
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "hipporag_amd", "csrc", KERNEL_SOURCES.get(kernel, ""))
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16] if os.path.isfile(path) else None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "cfg3:B256"   # what tools/pmc_target.py ran
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0], newline="")))
        ours = [r for r in rows if "hrag::" in r["Name"]]
        others = [r for r in rows if "hrag::" not in r["Name"]][:5]
        with open(dst + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in ours + others:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.0f}',
                            r["Percentage"], r["MinNs"], r["MaxNs"]])
    pmc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            if "hrag::" in r["Kernel_Name"]:
                pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in pmc.items():
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d["launches"] = {c: len(v) for c, v in cs.items()}
        if "FETCH_SIZE" in d:
            d["fetch_bytes_corrected"] = 2.0 * d["FETCH_SIZE"] * 1024.0
        if "WRITE_SIZE" in d:
            d["write_bytes"] = d["WRITE_SIZE"] * 1024.0
        if "fetch_bytes_corrected" in d and "write_bytes" in d:
            d["bytes_per_launch"] = d["fetch_bytes_corrected"] + d["write_bytes"]
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        out[k] = d
    if out:
        json.dump(out, open(dst + "_pmc.json", "w"), indent=1, sort_keys=True)
        # what bench.py's roofline.traffic reads: HBM bytes per launch of each SpMM kernel, latest run
        traffic, seen = {}, {}
        inst8 = {}    # ppr8_kernel<mode, residual form> -> HBM bytes per launch (bench.py weights them by the stage plan)
        for k, d in out.items():
            base = k.split("::")[-1].split("<")[0]
            if base == "ppr8_pair_kernel":      # two slabs per wavefront: the same sweep, the usual kernel at B >= 256
                base = "ppr8_kernel"
            if base == "ppr8_kernel" and "bytes_per_launch" in d and "<" in k:
                # <mode, residual form, est>: the third argument marks the instantiation that also measures the
                # convergence contract's est (checkpoint boundary / final sweep) -> "<m,r>" or "<m,r,est>"
                targs = k[k.index("<") + 1:k.rindex(">")].replace(" ", "").split(",")
                key = "<" + ",".join(targs[:2]) + (",est" if len(targs) > 2 and targs[2] == "true" else "") + ">"
                inst8[key] = {"bytes_per_launch": d["bytes_per_launch"], "l2_hit_rate": d.get("l2_hit_rate")}
            n = d["launches"].get("FETCH_SIZE", 0)
            if "bytes_per_launch" in d and base in ("ppr8_kernel", "ppr16_kernel", "ppr_spmm_kernel") and n > seen.get(base, 0):
                seen[base] = n      # the variant with the most launches (mode H for ppr16_kernel, C for ppr8_kernel)
                traffic[base] = {"bytes_per_launch": d["bytes_per_launch"], "l2_hit_rate": d.get("l2_hit_rate"),
                                 "workload": workload, "source": os.path.basename(dst) + "_pmc.json",
                                 # bench.py replays these figures only while the kernel's source is the one they were
                                 # collected on (a changed kernel makes roofline.traffic null until the passes are re-run)
                                 "kernel_source_sha16": kernel_source_sha16(base)}
        if inst8 and "ppr8_kernel" in traffic:
            traffic["ppr8_kernel"]["by_instantiation"] = inst8
        if traffic:
            tp = os.path.join(os.path.dirname(dst) or ".", "pmc_traffic.json")
            old = json.load(open(tp)) if os.path.exists(tp) else {}
            old.update(traffic)
            json.dump(old, open(tp, "w"), indent=1, sort_keys=True)
    print("wrote", dst + "_kernel_stats.csv" if stats else "(no stats)", dst + "_pmc.json" if out else "(no pmc)")


if __name__ == "__main__":
    main()
