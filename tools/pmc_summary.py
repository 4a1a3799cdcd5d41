#!/usr/bin/env python
"""Summarise rocprofv3 --pmc output directories into HBM bytes per launch of the dominant kernel.

    python tools/pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [kernel substring]

Corrections as prescribed by /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of 16-B-per-lane
reads (TCC_EA0_RDREQ x 64 B for 128-B requests) -> doubled; WRITE_SIZE is taken as reported
(uncalibrated).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(d, counter):
    vals = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True):
        with open(path, newline="") as f:
            rd = csv.DictReader(f)
            cols = rd.fieldnames or []
            if "Counter_Name" not in cols:
                continue
            for row in rd:
                if row["Counter_Name"] == counter:
                    vals[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return vals


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    sub = sys.argv[3] if len(sys.argv) > 3 else "ppr_spmm_kernel"
    out = {"kernel": sub, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)"}
    for name, d, counter in (("fetch", fetch_dir, "FETCH_SIZE"), ("write", write_dir, "WRITE_SIZE")):
        vals = collect(d, counter)
        hits = [v for k, vs in vals.items() if sub in k for v in vs]
        out[f"{name}_kib_per_launch_raw"] = sum(hits) / len(hits) if hits else None
        out[f"{name}_launches"] = len(hits)
    f, w = out["fetch_kib_per_launch_raw"], out["write_kib_per_launch_raw"]
    if f is not None and w is not None:
        out["fetch_bytes_corrected"] = 2.0 * f * 1024.0
        out["write_bytes"] = w * 1024.0
        out["bytes_per_launch"] = out["fetch_bytes_corrected"] + out["write_bytes"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
