#!/usr/bin/env python3
"""Reduce a rocprofv3 kernel trace (…_kernel_trace.csv) to device-occupancy figures for the
scoring loop: per hardware queue and overall, the busy time (union of kernel intervals), the
span, the number of kernels and the gaps between consecutive kernels; restricted to the window
between the first and last launch of the K1 kernel (the timed region of bench.py).

    python3 tools/trace_busy.py /tmp/prof/*/*_kernel_trace.csv > profiles/rNN_busy.json
"""
import csv
import json
import sys


def union(intervals):
    intervals.sort()
    busy, cur_s, cur_e = 0, None, None
    for s, e in intervals:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"],
                         r["Kernel_Name"]))
    k1 = [r for r in rows if "zo_perturb_units_kernel" in r[3]]
    if len(k1) >= 4:                      # skip the warm-up launches: window of the last 3/4
        k1.sort()
        lo, hi = k1[len(k1) // 4][0], max(r[1] for r in rows)
        rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    span = max(r[1] for r in rows) - min(r[0] for r in rows)
    out = {"window_ms": span / 1e6, "kernels": len(rows),
           "busy_any_queue_frac": union([(r[0], r[1]) for r in rows]) / span,
           "sum_kernel_time_over_window": sum(r[1] - r[0] for r in rows) / span, "queues": {}}
    by_q = {}
    for r in rows:
        by_q.setdefault(r[2], []).append(r)
    for q, rs in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
        rs.sort()
        gaps = [max(0, rs[i + 1][0] - rs[i][1]) for i in range(len(rs) - 1)]
        gaps.sort()
        durs = sorted(r[1] - r[0] for r in rs)
        out["queues"][q] = {
            "kernels": len(rs), "busy_frac_of_window": union([(r[0], r[1]) for r in rs]) / span,
            "median_kernel_us": durs[len(durs) // 2] / 1e3, "mean_kernel_us": sum(durs) / len(durs) / 1e3,
            "median_gap_us": gaps[len(gaps) // 2] / 1e3 if gaps else None,
            "mean_gap_us": sum(gaps) / len(gaps) / 1e3 if gaps else None,
            "gap_p90_us": gaps[int(0.9 * len(gaps))] / 1e3 if gaps else None}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
