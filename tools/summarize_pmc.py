#!/usr/bin/env python3
"""Turn rocprofv3 --pmc CSV output (counter_collection.csv) of tools/k1_launches.py into
HBM bytes per launch, with the gfx950 corrections of MI355X_MICROARCH.md §HBM:
FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced read stream ->
doubled; WRITE_SIZE is exact for 16 B/lane streaming stores; both are in KiB units.

    python3 tools/summarize_pmc.py <fetch_csv> <write_csv> <launches_json> [<existing summary to merge into>] > out.json
"""
import csv
import json
import sys


def per_kernel(path, counter, kernel_substr):
    vals = []
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter and kernel_substr in r.get("Kernel_Name", ""):
            vals.append(float(r["Counter_Value"]))
    return vals


def main():
    fetch_csv, write_csv, launches_json = sys.argv[1:4]
    doc = json.load(open(launches_json))
    launches = doc["launches"]
    form = doc.get("form", "units")
    kernel = {"units": "zo_perturb_units_kernel", "block": "zo_perturb_layers_kernel",
              "torch_block": "zo_torch_layers_kernel"}[form]
    f = per_kernel(fetch_csv, "FETCH_SIZE", kernel)
    w = per_kernel(write_csv, "WRITE_SIZE", kernel)
    n = min(len(f), len(w), len(launches))
    by_shape = {}
    for i in range(n):
        L = launches[i]
        d = by_shape.setdefault((form + ":" if form == "torch_block" else "") + L["shape"], {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0,
                                             "algorithmic_bytes": L["algorithmic_bytes"]})
        d["launches"] += 1
        d["fetch_kib"] += f[i]
        d["write_kib"] += w[i]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
                     "tools/k1_launches.py; FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "shapes": {}}
    tot_t = tot_a = 0.0
    for k, d in by_shape.items():
        traffic = (2 * d["fetch_kib"] + d["write_kib"]) * 1024 / d["launches"]
        out["shapes"][k] = {"launches": d["launches"], "hbm_bytes_per_launch": traffic,
                            "raw_fetch_kib_per_launch": d["fetch_kib"] / d["launches"],
                            "raw_write_kib_per_launch": d["write_kib"] / d["launches"],
                            "algorithmic_bytes_per_launch": d["algorithmic_bytes"],
                            "traffic_over_algorithmic": traffic / d["algorithmic_bytes"]}
        tot_t += traffic * d["launches"]
        tot_a += d["algorithmic_bytes"] * d["launches"]
    out[form] = {"traffic_over_algorithmic": tot_t / tot_a}
    if len(sys.argv) > 4:          # merge into an existing summary (other kernel forms are kept)
        old = json.load(open(sys.argv[4]))
        old.setdefault("shapes", {}).update(out["shapes"])
        old[form] = out[form]
        out = old
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
