#!/usr/bin/env python3
"""Per-shape summary of the layer-batched K1 launches in a rocprofv3 --kernel-trace CSV.

    python3 tools/k1_trace_summary.py <kernel_trace.csv> [--units 16] [--skip N] [--out k1_trace.csv]

The grid tells the matrix (one 64-thread workgroup per 1024 16-bit elements, rounded up to a
multiple of 8); algorithmic bytes = (2*U + 2) * 2 * numel.  --skip drops the
first N K1 launches (the bench's untimed warm-up)."""
import argparse
import csv
import json

SHAPES = {5120 * 2048: "t5_wi_wo 5120x2048", 2048 * 2048: "t5_qkvo 2048x2048",
          6144 * 1408: "vit_fc 6144x1408", 4224 * 1408: "vit_qkv 4224x1408",
          1408 * 1408: "vit_proj 1408x1408"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--units", type=int, default=16)
    ap.add_argument("--skip", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--bench-json", default=None,
                    help="the JSON line bench.py printed in the same run: its roofline.per_launch "
                         "bytes are matched to the K1 dispatches in order (needed for the "
                         "block-batched kernel, whose grid does not name one matrix)")
    args = ap.parse_args()
    by_grid = {}
    for numel, name in SHAPES.items():
        wgs = -(-(numel // 8) // 128)
        wgs = -(-wgs // 8) * 8
        by_grid[wgs * 64] = (name, numel)
    rows, others = [], []
    with open(args.trace) as f:
        rd = csv.DictReader(f)
        fields = rd.fieldnames
        for r in rd:
            if ("zo_perturb_units_kernel" in r["Kernel_Name"] or "zo_perturb_layers_kernel" in r["Kernel_Name"]
                    or "zo_torch_layers_kernel" in r["Kernel_Name"]):
                rows.append(r)
            else:
                others.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[args.skip:]
    if args.out:
        with open(args.out, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=fields, quoting=csv.QUOTE_NONNUMERIC)
            w.writeheader()
            w.writerows(rows)
    per_launch = None
    if args.bench_json:
        line = [ln for ln in open(args.bench_json).read().splitlines() if ln.startswith("{")][-1]
        per_launch = json.loads(line)["roofline"]["per_launch"]
        rows = rows[len(rows) - len(per_launch):]          # the timed region's launches
    # what else was on the device while each K1 launch ran (K1 shares HBM with it)
    for r in rows:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        hit = {}
        for s0, s1, nm in others:
            ov = min(b, s1) - max(a, s0)
            if ov > 0:
                hit[nm] = hit.get(nm, 0) + ov
        tot = sum(hit.values())
        top = sorted(hit.items(), key=lambda kv: -kv[1])[:3]
        print(f"K1 launch {(b - a) / 1e3:8.2f} us: other kernels overlapping it {tot / 1e3:8.2f} us"
              + "".join(f"  [{nm} {v / 1e3:.1f}]" for nm, v in top))
    agg, tot_b, tot_t = {}, 0.0, 0.0
    for i, r in enumerate(rows):
        g = int(r["Grid_Size_X"])
        name, numel = by_grid.get(g, (f"grid {g}", None))
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        b = (2 * args.units + 2) * 2 * numel if numel else 0
        if per_launch is not None:
            b = per_launch[i]["bytes"]
            if numel is None:
                name = f"block launch, {b / 1e6:.0f} MB"
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += us
        a[2] += b
        tot_b += b
        tot_t += us
    out = {"launches": len(rows), "units": args.units, "shapes": {}}
    for name, (n, us, b) in sorted(agg.items()):
        out["shapes"][name] = {"launches": n, "avg_us": us / n, "gbs": b / us / 1e3,
                               "frac_of_8TBs": b / us / 1e3 / 8000}
        print(f"{name:22s} x{n:3d}  avg {us / n:8.2f} us  {b / us / 1e3:6.0f} GB/s  {b / us / 80e3:5.1f} %")
    if tot_t:
        out["all"] = {"avg_us": tot_t / len(rows), "gbs": tot_b / tot_t / 1e3,
                      "frac_of_8TBs": tot_b / tot_t / 1e3 / 8000}
        print(f"all launches: sum bytes / sum time = {tot_b / tot_t / 1e3:.0f} GB/s = "
              f"{tot_b / tot_t / 80e3:.1f} % of 8 TB/s, avg {tot_t / len(rows):.2f} us")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
