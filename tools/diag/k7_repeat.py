#!/usr/bin/env python3
"""The same ViT-g block (4 fp16 matrices) through the K7 block call 300 times, other work on a second
stream in between: every result equal to the first, and to the reference's expression."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ecoflap_amd import hip  # noqa: E402


def main():
    kern = hip.HipKernels()
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(4224, 1408), (1408, 1408), (6144, 1408), (1408, 6144)]
    ws = [(torch.randn(r, c, device="cuda", generator=g) * 0.02).half() for r, c in shapes]
    srs = [torch.rand(c, device="cuda", generator=g) + 0.05 for _, c in shapes]
    ks = [r * c // 2 for r, c in shapes]
    want = []
    for w, s, k in zip(ws, srs, ks):
        m = w.abs().float() * torch.sqrt(s).reshape(1, -1)
        t = torch.sort(m.flatten())[0][k]
        want.append(torch.where(m <= t, torch.zeros_like(w), w))
    side = torch.cuda.Stream()
    x = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
    bad = 0
    for rep in range(300):
        cur = [w.clone() for w in ws]
        if rep % 2:
            with torch.cuda.stream(side):           # something else on the device at the same time
                for _ in range(3):
                    x @ x
        kern.wanda_prune_block([(w, s, "matrix", k, None) for w, s, k in zip(cur, srs, ks)])
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(cur, want))
    print(f"300 repetitions x 4 matrices: {bad} results differ from sort(metric)[k]; fallbacks {kern.wanda_fallback_counts()}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
