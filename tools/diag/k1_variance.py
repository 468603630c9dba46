#!/usr/bin/env python3
"""Why does the same block-batched K1 launch (ViT-g block: qkv, proj, fc1, fc2 x 16 units,
1.7 GB) take 277 us in one bench run and 355 us in another, with nothing else on the device
(tools/k1_trace_summary.py prints the overlap: 0) and the same 1 GiB fill rate?

Times the launch, in ONE process, (a) 40 times on the same buffers, (b) on fresh buffers from
the caching allocator each time (the loop's way), (c) on buffers carved from ONE arena
allocated up front, (d) after the device has been kept busy by GEMMs for a second (clocks /
temperature), and prints min / median / max per case.  Run it a few times: differences
between processes with (a) stable inside each point at physical placement or the box."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ecoflap_amd import hip  # noqa: E402

SHAPES = [(4224, 1408), (1408, 1408), (6144, 1408), (1408, 6144)]
U = 16


def build(kern, ws, alloc):
    batch, nbytes = [], 0
    for li, w in enumerate(ws):
        scr = alloc((2 * U,) + tuple(w.shape))
        fin = alloc(tuple(w.shape))
        batch.append((w, fin, list(range(100 * li, 100 * li + U)),
                      [scr[2 * i] for i in range(U)], [scr[2 * i + 1] for i in range(U)]))
        nbytes += (2 * U + 2) * 2 * w.numel()
    return batch, nbytes


def time_launch(kern, batch):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    kern.zo_perturb_layers(batch, 1e-3)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def stats(xs, nbytes):
    med = statistics.median(xs)
    return {"min_us": round(min(xs), 1), "median_us": round(med, 1), "max_us": round(max(xs), 1),
            "median_frac_of_8TBs": round(nbytes / med / 1e3 / 8000, 3)}


def main():
    kern = hip.HipKernels()
    ws = [(torch.randn(r, c, device="cuda") * 0.02).half() for r, c in SHAPES]
    plain = lambda shape: torch.empty(shape, dtype=torch.float16, device="cuda")   # noqa: E731
    out = {}
    batch, nbytes = build(kern, ws, plain)
    time_launch(kern, batch)
    out["same_buffers"] = stats([time_launch(kern, batch) for _ in range(40)], nbytes)
    xs = []
    for i in range(12):
        keep = batch                       # hold the old ones: the allocator must hand out new blocks
        batch, _ = build(kern, ws, plain)
        xs.append(time_launch(kern, batch))
        del keep
    out["fresh_buffers_each_launch_first_touch"] = stats(xs, nbytes)
    xs = [time_launch(kern, batch) for _ in range(10)]
    out["last_fresh_buffers_again"] = stats(xs, nbytes)
    # one arena up front
    total = sum((2 * U + 1) * w.numel() for w in ws) + 4096 * len(ws) * 2
    arena = torch.empty(total, dtype=torch.float16, device="cuda")
    cur = [0]

    def carve(shape):
        n = 1
        for d in shape:
            n *= d
        a = cur[0]
        cur[0] = (a + n + 1023) // 1024 * 1024
        return arena[a:a + n].view(shape)

    batch_a, _ = build(kern, ws, carve)
    time_launch(kern, batch_a)
    out["one_arena"] = stats([time_launch(kern, batch_a) for _ in range(20)], nbytes)
    # after sustained MFMA load
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
    for _ in range(200):
        a @ a
    torch.cuda.synchronize()
    out["after_1s_of_gemm"] = stats([time_launch(kern, batch_a) for _ in range(20)], nbytes)
    buf = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    fills = []
    for i in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); buf.fill_(i); e1.record()
        torch.cuda.synchronize()
        fills.append((1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    out["fill_1GiB_gbs"] = round(statistics.median(fills), 0)
    out["bytes_per_launch"] = nbytes
    print("K1VAR " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
