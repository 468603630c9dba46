#!/usr/bin/env python3
"""Fuzz of the K7 matrix-mode selection against the reference's expression on the GPU
(`thres = sort(metric.flatten())[k]; metric <= thres`, wanda_pruner.py:555-558): random shapes
around the ViT sizes, three dtypes, k from 0 to numel - 1, weight distributions with ties, zeros,
heavy tails and constant columns.    python3 tools/diag/k7_fuzz.py [cases] [seed]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ecoflap_amd import hip  # noqa: E402


def make(g, rows, cols, dt, kind):
    w = torch.randn(rows, cols, device="cuda", generator=g) * 0.02
    s = torch.rand(cols, device="cuda", generator=g) + 0.05
    if kind == "ties":
        w = torch.round(w * 50) / 50
    elif kind == "zeros":
        w = w * (torch.rand(rows, cols, device="cuda", generator=g) < 0.4)
    elif kind == "heavy":
        w = w * torch.exp(2.5 * torch.randn(rows, cols, device="cuda", generator=g))
    elif kind == "flat_scaler":
        s = torch.full((cols,), 0.3, device="cuda")
    elif kind == "quantised":
        w = torch.round(w * 400) / 400
        s = torch.full((cols,), 1.0, device="cuda")
    return w.to(dt), s


def one(kern, g, case):
    rows = int(torch.randint(96, 6200, (1,), generator=g, device="cuda"))
    cols = 8 * int(torch.randint(64, 800, (1,), generator=g, device="cuda"))
    dt = [torch.float16, torch.bfloat16, torch.float32][case % 3]
    kind = ["normal", "ties", "zeros", "heavy", "flat_scaler", "quantised", "normal"][case % 7]
    w, s = make(g, rows, cols, dt, kind)
    numel = rows * cols
    frac = [0.5, 0.37, 0.9, 0.05, 0.6, 0.999, 0.001, 0.25][case % 8]
    k = min(numel - 1, max(0, int(numel * frac)))
    if case % 29 == 0:
        k = 0
    if case % 31 == 0:
        k = numel - 1
    metric = w.abs().float() * torch.sqrt(s).reshape(1, -1)
    thres = torch.sort(metric.flatten())[0][k]
    want = torch.where(metric <= thres, torch.zeros_like(w), w)
    del metric
    got = w.clone()
    kern.wanda_prune_matrix(got, s, k)
    ok = torch.equal(got, want)
    return ok, (rows, cols, str(dt), kind, k)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    kern = hip.HipKernels()
    g = torch.Generator(device="cuda").manual_seed(seed)
    kern.wanda_fallback_counts()
    bad = []
    for case in range(n):
        ok, info = one(kern, g, case)
        if not ok:
            bad.append(info)
    print(f"{n} cases, {len(bad)} differences; fallbacks (misses, crowded): {kern.wanda_fallback_counts()}")
    for b in bad[:10]:
        print("  DIFFERENT:", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
