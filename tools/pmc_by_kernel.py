#!/usr/bin/env python3
"""Mean of every counter and of the dispatch duration per (kernel, grid size) out of a rocprofv3
--pmc counter_collection.csv; with GRBM_GUI_ACTIVE present also the effective clock
(GRBM_GUI_ACTIVE / 8 XCDs / duration) and, with SQ_VALU_MFMA_BUSY_CYCLES, the MFMA pipes' share of
the SIMD cycles (busy cycles / (4 SIMDs x 256 CUs x shader cycles of the dispatch)).

    python3 tools/pmc_by_kernel.py <counter_collection.csv> [substring ...]
"""
import csv
import sys
from collections import defaultdict


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if subs and not any(s in name for s in subs):
            continue
        key = (name[:90], int(r["Grid_Size"]))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for key in sorted(acc):
        d = sorted(dur[key].values())
        us = d[len(d) // 2]
        c = {k: sum(v) / len(v) for k, v in acc[key].items()}
        line = f"{key[0]:90s} grid {key[1]:9d} n {len(d):4d} median {us:8.1f} us"
        cyc = None
        if "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            line += f"  clock {cyc / us / 1e3:5.2f} GHz"
        if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            line += f"  mfma busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc) * 100:5.1f} % of SIMD cycles"
        print(line)
        print("      " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))


main()
