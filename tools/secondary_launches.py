#!/usr/bin/env python3
"""Time the multi-tensor kernels next to K1 at BASELINE sizes (HIP events, a GEMM queued ahead of
each timed launch so host gaps stay outside the event pair) and print GB/s against the algorithmic
bytes of DESIGN.md §4:

  K3+K4  ecoflap_absprod_reduce_multi   FlanT5-XL (432 matrices, bf16 W and g) and BLIP-2 (588)
  Real-* ecoflap_grad_accum_multi, ecoflap_global_threshold_prune, ecoflap_count_zeros_multi
  K8     ecoflap_mask_mul

Also the workload for the rocprofv3 summaries under profiles/."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecoflap_amd import hip  # noqa: E402


def shapes(kind):
    vit = [(4224, 1408), (1408, 1408), (6144, 1408), (1408, 6144)]
    enc = [(2048, 2048)] * 4 + [(5120, 2048)] * 2 + [(2048, 5120)]
    dec = [(2048, 2048)] * 8 + [(5120, 2048)] * 2 + [(2048, 5120)]
    out = []
    if kind == "blip2":
        out += [(s, torch.float16) for _ in range(39) for s in vit]
    out += [(s, torch.bfloat16) for _ in range(24) for s in enc]
    out += [(s, torch.bfloat16) for _ in range(24) for s in dec]
    return out


def timed(fn, reps=5):
    blocker = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        for _ in range(4):
            blocker @ blocker
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) * 1e3)
    return statistics.median(out), min(out)


def main():
    kern = hip.HipKernels()
    res = []
    for kind in ("t5", "blip2"):
        sh = shapes(kind)
        ws = [(torch.randn(s, device="cuda") * 0.02).to(dt) for s, dt in sh]
        gs = [(torch.randn(s, device="cuda") * 0.01).to(dt) for s, dt in sh]
        numel = sum(w.numel() for w in ws)
        sums = torch.zeros(len(ws), dtype=torch.float64, device="cuda")
        med, mn = timed(lambda: kern.absprod_reduce_pairs(ws, gs, hip.RED_ABSW_ABSG, sums))
        res.append((f"K3+K4 absprod_reduce_multi {kind} ({len(ws)} matrices)", 4 * numel, med, mn))
        if kind == "t5":
            continue
        accs = [torch.zeros(w.shape, dtype=torch.float32, device="cuda") for w in ws]
        med, mn = timed(lambda: kern.grad_accum_multi(accs, gs))
        res.append(("Real-* grad_accum_multi blip2 (acc rw + g)", (8 + 2) * numel, med, mn))
        masks = [torch.ones(w.shape, dtype=torch.uint8, device="cuda") for w in ws]
        # threshold at rank 1: nothing but the smallest score is pruned, so repeats see the same data
        med, mn = timed(lambda: kern.global_threshold_prune(ws, accs, masks, 0, 3.0, 1), reps=3)
        res.append(("Real-* global_threshold_prune blip2 (3 hist passes + apply)",
                    (3 * 7 + 10) * numel, med, mn))
        med, mn = timed(lambda: kern.count_zeros_multi(ws))
        res.append(("Real-* count_zeros_multi blip2", 2 * numel, med, mn))
        big = ws[2]
        km = (torch.rand(big.shape, device="cuda") > 0.5).to(torch.uint8)
        g = gs[2].clone()
        med, mn = timed(lambda: kern.mask_mul(g, km))
        res.append(("K8 mask_mul 6144x1408 fp16", (2 * 2 + 1) * big.numel(), med, mn))
    for name, nbytes, med, mn in res:
        print(f"{name:64s} {nbytes/1e9:7.2f} GB  median {med:9.1f} us  min {mn:9.1f} us  "
              f"{nbytes/med/1e3:7.0f} GB/s ({nbytes/med/1e3/80:5.1f}% of 8 TB/s)")
    print(json.dumps([{"kernel": n, "bytes": b, "median_us": m, "min_us": mn} for n, b, m, mn in res]))


if __name__ == "__main__":
    main()
