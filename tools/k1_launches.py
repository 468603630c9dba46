#!/usr/bin/env python3
"""Issue the layer-batched K1 kernel on BLIP-2's matrix shapes — the workload profiled
with rocprofv3 (--kernel-trace --stats for durations; separate --pmc FETCH_SIZE and
--pmc WRITE_SIZE passes for HBM traffic; see profiles/README.md).

    python3 tools/k1_launches.py [--units 16] [--reps 4]
Each launch touches a different weight/scratch set, cycling through > 1 GiB so that no
launch finds its matrix in the 256 MiB Infinity Cache (as in the real loop, where two
7.4 GB forwards separate launches on the same matrix).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecoflap_amd import hip  # noqa: E402

SHAPES = [("t5_wi_wo", 5120 * 2048, torch.bfloat16), ("t5_qkvo", 2048 * 2048, torch.bfloat16),
          ("vit_fc", 6144 * 1408, torch.float16), ("vit_qkv", 4224 * 1408, torch.float16),
          ("vit_proj", 1408 * 1408, torch.float16)]


BLOCKS = [("t5_block", [2048 * 2048] * 4 + [5120 * 2048] * 3, torch.bfloat16),
          ("vit_block", [4224 * 1408, 1408 * 1408, 6144 * 1408, 6144 * 1408], torch.float16)]


def block_form(kern, U, reps, z=None):
    """One launch per transformer block: every matrix of the block with its U units (the
    default form of the stage-1 loop).  Two buffer sets per block shape, > 1 GiB each."""
    plan = []
    for name, numels, dt in BLOCKS:
        sets = []
        for _ in range(2):
            ws = [torch.randn(n, device="cuda").mul_(0.02).to(dt) for n in numels]
            fin = [torch.empty_like(w) for w in ws]
            scr = [torch.empty(2 * U, n, device="cuda", dtype=dt) for n in numels]
            sets.append((ws, fin, scr))
        plan.append((name, numels, dt, sets))
    torch.cuda.synchronize()
    launches = []
    for rep in range(reps):
        for name, numels, dt, sets in plan:
            for k, (ws, fin, scr) in enumerate(sets):
                layers = [(w, f, [1000 * rep + 100 * k + 16 * i + u for u in range(U)],
                           [s[2 * u] for u in range(U)], [s[2 * u + 1] for u in range(U)])
                          + ((z,) if z else ())
                          for i, (w, f, s) in enumerate(zip(ws, fin, scr))]
                kern.zo_perturb_layers(layers, 1e-3)
                launches.append({"shape": name, "numel": sum(numels), "dtype": str(dt),
                                 "algorithmic_bytes": (2 * U + 2) * 2 * sum(numels)})
    torch.cuda.synchronize()
    print(json.dumps({"units": U, "form": "torch_block" if z else "block", "launches": launches}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--units", type=int, default=16)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--only", default=None)
    ap.add_argument("--form", choices=["units", "block"], default="units",
                    help="block: one zo_perturb_layers launch per transformer block (all of its "
                         "matrices), as the stage-1 loop issues it by default")
    ap.add_argument("--z", choices=["philox", "torch"], default="philox",
                    help="torch: the reference's draw regenerated in registers (zo_torch_layers_kernel, the "
                         "default mode of the loop since round 5); block form only")
    args = ap.parse_args()
    kern = hip.HipKernels()
    U = args.units
    if args.form == "block":
        return block_form(kern, U, args.reps, hip.TORCH_Z if args.z == "torch" else None)
    plan = []
    for name, n, dt in SHAPES:
        if args.only and args.only != name:
            continue
        sets = max(2, int(1.2e9 // ((2 * U + 1) * n * 2)) + 1)
        ws = [torch.randn(n, device="cuda").mul_(0.02).to(dt) for _ in range(sets)]
        scr = [torch.empty(2 * U, n, device="cuda", dtype=dt) for _ in range(sets)]
        plan.append((name, n, dt, ws, scr))
    torch.cuda.synchronize()
    launches = []
    for rep in range(args.reps):
        for name, n, dt, ws, scr in plan:
            for k in range(len(ws)):
                kern.zo_perturb_units(ws[k], 1e-3, [1000 * rep + 16 * k + u for u in range(U)],
                                      [scr[k][2 * u] for u in range(U)],
                                      [scr[k][2 * u + 1] for u in range(U)])
                launches.append({"shape": name, "numel": n, "dtype": str(dt),
                                 "algorithmic_bytes": (2 * U + 2) * 2 * n})
    torch.cuda.synchronize()
    print(json.dumps({"units": U, "launches": launches}))


if __name__ == "__main__":
    main()
