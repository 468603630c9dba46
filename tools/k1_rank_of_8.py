#!/usr/bin/env python3
"""What K1 costs one rank of a data-parallel world of 8 (weak scaling: 128 units per matrix, 16
owned): every rank chains ALL units' perturbations through W (the drifted weights must equal
the one-process run's, SURVEY.md 8e iii) but writes theta+/theta- only for its own.  Measured on
one GPU for the five BLIP-2 matrix shapes: one world-1 launch (16 of 16 owned) next to the
world-8 form (4 launches of 32 units, 4 owned each)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecoflap_amd import hip  # noqa: E402

SHAPES = [("t5.wi/wo", 5120, 2048, torch.bfloat16), ("t5.qkvo", 2048, 2048, torch.bfloat16),
          ("vit.fc", 6144, 1408, torch.float16), ("vit.qkv", 4224, 1408, torch.float16),
          ("vit.proj", 1408, 1408, torch.float16)]


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    kern = hip.HipKernels()
    rows = []
    for name, r, c, dt in SHAPES:
        w = (torch.randn(r, c, device="cuda") * 0.02).to(dt)
        scratch = torch.empty(32, r, c, dtype=dt, device="cuda")
        seeds = list(range(1000, 1128))

        def world1():
            kern.zo_perturb_units(w, 1e-3, seeds[:16], [scratch[2 * i] for i in range(16)],
                                  [scratch[2 * i + 1] for i in range(16)])

        def world8(rank=3):
            own = [(u % 8) == rank for u in range(128)]
            k = 0
            plus, minus = [], []
            for mine in own:
                plus.append(scratch[2 * k] if mine else None)
                minus.append(scratch[2 * k + 1] if mine else None)
                k += int(mine)
            kern.zo_perturb_units(w, 1e-3, seeds, plus, minus)

        t1, t8 = timed(world1), timed(world8)
        rows.append({"matrix": name, "numel": r * c, "world1_16_units_us": round(t1, 1),
                     "world8_128_units_16_owned_us": round(t8, 1),
                     "extra_us_per_matrix": round(t8 - t1, 1)})
    total_extra = sum(x["extra_us_per_matrix"] * n for x, n in zip(rows, (144, 288, 78, 39, 39)))
    print(json.dumps({"per_matrix": rows, "extra_ms_per_588_matrix_pass": round(total_extra / 1e3, 1)}))


if __name__ == "__main__":
    main()
