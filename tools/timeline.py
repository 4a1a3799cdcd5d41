#!/usr/bin/env python
"""Timeline of the LAST retrieve in a rocprofv3 kernel trace (tools/trace_target.py synchronises after every
call, so calls are separated by the largest gaps):  python tools/timeline.py <kernel_trace.csv> [calls]"""
import csv
import re
import sys


def main():
    path, calls = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cut = max([i for i in range(1, len(rows)) if rows[i][0] - rows[i - 1][1] > 100_000] or [0])   # > 100 us idle
    last = rows[cut:]
    t0 = last[0][0]
    busy = 0
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7}  kernel")
    prev_end = t0
    for s, e, name in last:
        short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {short}")
        busy += e - s
        prev_end = e
    span = last[-1][1] - t0
    print(f"kernels {len(last)}  span {span / 1e3:.1f} us  busy {busy / 1e3:.1f} us  gaps {(span - busy) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
