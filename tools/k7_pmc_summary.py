#!/usr/bin/env python3
"""HBM bytes per K7 matrix-mode block call out of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
over `tools/wanda_launches.py --only matrixblock`, with the gfx950 corrections of
MI355X_MICROARCH.md §HBM (KiB units; FETCH_SIZE reports half the bytes of a 16 B/lane read stream ->
doubled; WRITE_SIZE exact), against the algorithmic 2*s*numel + 4*cols of the ViT-g block.

    python3 tools/k7_pmc_summary.py <fetch_csv> <write_csv>
"""
import csv
import json
import sys
from collections import defaultdict

SHAPES = [(4224, 1408), (1408, 1408), (6144, 1408), (1408, 6144)]
ALGORITHMIC = sum(2 * 2 * r * c + 4 * c for r, c in SHAPES)
PATHS = {"sampled (default)": ("wanda_matrix_sbracket_kernel", "wanda_matrix_apply2_kernel"),
         "three histograms (ECOFLAP_WANDA_SAMPLED=0)": ("wanda_matrix_hist_kernel", "wanda_matrix_apply_kernel",
                                                        "sqrt_cols_kernel")}


def read(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("<")[0].replace("void ", "")
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return tot, n


def main():
    f, fn = read(sys.argv[1], "FETCH_SIZE")
    w, wn = read(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/wanda_launches.py "
                     "--only matrixblock; FETCH_SIZE doubled per MI355X_MICROARCH.md; KiB units",
           "algorithmic_bytes_per_call": ALGORITHMIC, "paths": {}}
    for name, kernels in PATHS.items():
        calls = min(fn.get(kernels[1], 0), wn.get(kernels[1], 0))     # one apply launch per call
        if not calls:
            continue
        per = {}
        traffic = 0.0
        for k in kernels:
            if k not in fn:
                continue
            fb = 2 * f[k] * 1024 / calls
            wb = w.get(k, 0.0) * 1024 / calls
            per[k] = {"launches_per_call": fn[k] / calls, "fetch_bytes_per_call": fb, "write_bytes_per_call": wb}
            traffic += fb + wb
        out["paths"][name] = {"calls": calls, "kernels": per, "hbm_bytes_per_call": traffic,
                              "traffic_over_algorithmic": traffic / ALGORITHMIC}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
