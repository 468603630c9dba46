#!/usr/bin/env python3
"""K1 in its default mode (torch's draw regenerated in registers, zo_torch_layers_kernel) outside
the scoring loop: the bench's block launches on random weights, timed by torch events on the launch
stream, in several surroundings — to tell what a launch owes to its shape from what it owes to its
place in the loop (profiles/r06_k1/).

    python3 tools/diag/k1_torch_standalone.py [--units 16] [--reps 6]
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ecoflap_amd import hip  # noqa: E402
from bench import HipEvents  # noqa: E402  (raw hipEvent_t pairs filled by the launch itself)

Q, WI = 2048 * 2048, 5120 * 2048
CASES = [
    ("vit_block f16 (qkv proj fc1 fc2)", torch.float16, [4224 * 1408, 1408 * 1408, 6144 * 1408, 6144 * 1408]),
    ("vit_block shapes as bf16", torch.bfloat16, [4224 * 1408, 1408 * 1408, 6144 * 1408, 6144 * 1408]),
    ("vit fc1 alone f16", torch.float16, [6144 * 1408]),
    ("vit qkv alone f16", torch.float16, [4224 * 1408]),
    ("t5 4xqkvo + 2xwi bf16", torch.bfloat16, [Q, Q, Q, Q, WI, WI]),
    ("t5 shapes as f16", torch.float16, [Q, Q, Q, Q, WI, WI]),
    ("t5 lone pair 2xqkvo bf16", torch.bfloat16, [Q, Q]),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--units", type=int, default=16)
    ap.add_argument("--reps", type=int, default=6)
    args = ap.parse_args()
    kern = hip.HipKernels()
    hev = HipEvents()
    U = args.units
    dev = torch.device("cuda")
    a = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    out = {}
    for name, dt, numels in CASES:
        sets = []
        for _ in range(2):
            ws = [torch.randn(n, device=dev).mul_(0.02).to(dt) for n in numels]
            fin = [torch.empty_like(w) for w in ws]
            scr = [torch.empty(2 * U, n, device=dev, dtype=dt) for n in numels]
            sets.append((ws, fin, scr))
        nbytes = (2 * U + 2) * 2 * sum(numels)
        for surround in ("idle", "after_gemms", "in_place"):
            us = []
            for rep in range(args.reps + 2):
                ws, fin, scr = sets[rep & 1]
                layers = [(w, (w if surround == "in_place" else f), [1000 * rep + 16 * i + u for u in range(U)],
                           [s[2 * u] for u in range(U)], [s[2 * u + 1] for u in range(U)])
                          for i, (w, f, s) in enumerate(zip(ws, fin, scr))]
                if surround == "after_gemms":
                    for _ in range(6):
                        a @ a
                pair = hev.pair()
                kern.zo_perturb_layers_torch(layers, 1e-3, events=lambda: pair)
                t = hev.elapsed_us(*pair)       # the kernel's own begin / end timestamps
                if rep >= 2:
                    us.append(t)
            med = statistics.median(us)
            out[f"{name} | {surround}"] = {"median_us": med, "min_us": min(us), "frac_of_8TBs_median": nbytes / med / 8e6,
                                           "frac_of_8TBs_best": nbytes / min(us) / 8e6, "MB": nbytes / 1e6}
            print(f"{name:36s} {surround:12s} median {med:8.1f} us  min {min(us):8.1f}  "
                  f"{100 * nbytes / med / 8e6:5.1f} % (median)  {100 * nbytes / min(us) / 8e6:5.1f} % (best)", flush=True)
        del sets
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
