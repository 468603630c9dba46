#!/usr/bin/env python3
"""Sum rocprofv3 kernel-trace durations per call of a block-level operation: kernels are grouped
by launch order into calls (a call starts at the kernel named by --first), per-call totals are
printed with their median.   python3 kernel_groups.py trace.csv --first sqrt_cols --match wanda_matrix,sqrt_cols"""
import csv
import statistics
import sys


def main():
    path = sys.argv[1]
    first = sys.argv[sys.argv.index("--first") + 1]
    match = sys.argv[sys.argv.index("--match") + 1].split(",")
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    calls, cur = [], None
    for r in rows:
        n = r["Kernel_Name"]
        if not any(m in n for m in match):
            continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if first in n:
            cur = {"kernels": [], "t0": int(r["Start_Timestamp"])}
            calls.append(cur)
        if cur is not None:
            cur["kernels"].append((n.split("(")[0][-40:], round(d, 1)))
            cur["t1"] = int(r["End_Timestamp"])
    sig = {}
    for c in calls:
        key = tuple(k for k, _ in c["kernels"])
        sig.setdefault(key, []).append((sum(d for _, d in c["kernels"]), (c["t1"] - c["t0"]) / 1e3, c["kernels"]))
    for key, lst in sig.items():
        busy = [x[0] for x in lst]
        span = [x[1] for x in lst]
        print(f"{len(lst):4d} calls  kernel-time median {statistics.median(busy):8.1f} us  "
              f"first-start..last-end median {statistics.median(span):8.1f} us   {' + '.join(k for k in key)}")
        print("       last call:", lst[-1][2])


if __name__ == "__main__":
    main()
