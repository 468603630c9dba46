#!/usr/bin/env python3
"""Fuzz of the K7 rows-mode selection against the reference's expression on the GPU
(`sort_res = torch.sort(W_metric, dim=-1, stable=True); indices = sort_res[1][:, :k]; W[mask] = 0`,
wanda_pruner.py:260-279): random shapes around the FlanT5 sizes (wave form up to 2048 columns, workgroup
form beyond, odd widths through the LDS form), three dtypes, k from 0 to cols, ties / zeros / heavy tails.
    python3 tools/diag/k7_rows_fuzz.py [cases] [seed]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ecoflap_amd import hip  # noqa: E402


def one(kern, g, case):
    rows = int(torch.randint(1, 700, (1,), generator=g, device="cuda"))
    cols = int(torch.randint(8, 1030, (1,), generator=g, device="cuda"))
    cols = cols * 8 if case % 5 else cols + 3               # every fifth: an odd width
    dt = [torch.bfloat16, torch.float16, torch.float32][case % 3]
    kind = ["normal", "ties", "zeros", "heavy", "normal"][case % 5 if case % 5 else 0]
    w = torch.randn(rows, cols, device="cuda", generator=g) * 0.02
    s = torch.rand(cols, device="cuda", generator=g) + 0.05
    if kind == "ties":
        w = torch.round(w * 50) / 50
        s = torch.full((cols,), 0.25, device="cuda")
    elif kind == "zeros":
        w = w * (torch.rand(rows, cols, device="cuda", generator=g) < 0.4)
    elif kind == "heavy":
        w = w * torch.exp(2.5 * torch.randn(rows, cols, device="cuda", generator=g))
    w = w.to(dt)
    frac = [0.5, 0.37, 0.9, 0.05, 0.6, 1.0, 0.0, 0.25][case % 8]
    k = int(cols * frac)
    metric = w.abs().float() * torch.sqrt(s).reshape(1, -1)
    idx = torch.sort(metric, dim=-1, stable=True)[1][:, :k]
    mask = torch.zeros(rows, cols, dtype=torch.bool, device="cuda")
    mask.scatter_(1, idx, True)
    want = torch.where(mask, torch.zeros_like(w), w)
    got = w.clone()
    kern.wanda_prune_rows(got, s, k)
    return torch.equal(got, want), (rows, cols, str(dt), kind, k)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    kern = hip.HipKernels()
    g = torch.Generator(device="cuda").manual_seed(seed)
    bad = []
    for case in range(n):
        ok, info = one(kern, g, case)
        if not ok:
            bad.append(info)
    print(f"{n} cases, {len(bad)} differences")
    for b in bad[:10]:
        print("  DIFFERENT:", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
