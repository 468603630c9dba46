#!/usr/bin/env python3
"""Per-shape time of the layer-batched K1 kernel (HIP events on the launch stream, cold HBM:
launches cycle through > 1 GiB of weight / scratch sets; a short GEMM queued before each timed
launch keeps the host's enqueue gap out of the event pair).

    python3 tools/k1_time.py [--units 16] [--reps 3]        # ECOFLAP_HIP_LIB=... for A/B builds
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecoflap_amd import hip  # noqa: E402
from k1_launches import SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--units", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--blocker", default="gemm", choices=["gemm", "sleep", "none"],
                    help="what keeps the queue busy before each timed launch")
    ap.add_argument("--data", default="random", choices=["random", "const"])
    args = ap.parse_args()
    kern = hip.HipKernels()
    U = args.units
    blocker = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    out = {}
    tot_b = tot_t = 0.0
    for name, n, dt in SHAPES:
        sets = max(3, int(1.2e9 // ((2 * U + 1) * n * 2)) + 1)
        ws = [(torch.randn(n, device="cuda").mul_(0.02) if args.data == "random"
               else torch.full((n,), 0.0115, device="cuda")).to(dt) for _ in range(sets)]
        scr = [torch.empty(2 * U, n, device="cuda", dtype=dt) for _ in range(sets)]

        def launch(i):
            k = i % sets
            kern.zo_perturb_units(ws[k], 1e-3, [7 * i + u for u in range(U)],
                                  [scr[k][2 * u] for u in range(U)],
                                  [scr[k][2 * u + 1] for u in range(U)])
        for i in range(sets):
            launch(i)
        torch.cuda.synchronize()
        times = []
        for rep in range(args.reps):
            evs = []
            for i in range(sets):
                if args.blocker == "gemm":
                    for _ in range(3):
                        torch.mm(blocker, blocker)
                elif args.blocker == "sleep":
                    torch.cuda._sleep(600000)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                launch(i)
                e.record()
                evs.append((s, e))
            torch.cuda.synchronize()
            times += [s.elapsed_time(e) * 1e3 for s, e in evs]
        nbytes = (2 * U + 2) * 2 * n
        med = statistics.median(times)
        mean = statistics.mean(times)
        out[name] = {"dtype": str(dt), "numel": n, "bytes": nbytes, "median_us": med, "mean_us": mean,
                     "gbs_mean": nbytes / mean / 1e3, "frac_of_8TBs": nbytes / mean / 1e3 / 8000}
        tot_b += nbytes
        tot_t += mean
        print(f"{name:10s} {str(dt):15s} {nbytes / 1e6:7.1f} MB  median {med:7.2f} us  mean {mean:7.2f} us"
              f"  {nbytes / mean / 1e3:6.0f} GB/s  {nbytes / mean / 1e3 / 80:5.1f} %", flush=True)
        del ws, scr
    print(json.dumps({"units": U, "blocker": args.blocker, "data": args.data, "lib": os.environ.get("ECOFLAP_HIP_LIB", "in-tree"),
                      "all_shapes_gbs": tot_b / tot_t / 1e3, "shapes": out}))


if __name__ == "__main__":
    main()
