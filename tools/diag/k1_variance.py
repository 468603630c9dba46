#!/usr/bin/env python3
"""Why does the same block-batched K1 launch (ViT-g block: qkv, proj, fc1, fc2 x 16 units,
1.7 GB) take 277 us in one bench run and 355 us in another, with nothing else on the device
(tools/k1_trace_summary.py prints the overlap: 0) and the same 1 GiB fill rate?

Times the launch by its own begin / end timestamps, in ONE process, (a) 40 times on the same
buffers, (b) on fresh buffers from the caching allocator each time (the loop's way), (c) on
buffers carved from ONE arena allocated up front, (d) queued right behind 1 / 10 / 100 / 400
8192^3 fp16 GEMMs, (e) behind 100 GEMMs and a 0.3 / 3 ms near-idle gap, and prints min / median /
max per case.  Run it a few times: differences
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


_EV = None


def time_launch(kern, batch, before=None):
    """The kernel's own begin / end timestamps (HIP events attached to the launch, as bench.py's
    roofline leg does): nothing of the host's table building or the queue in front of it."""
    global _EV
    if _EV is None:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        from bench import HipEvents
        _EV = HipEvents()
    pair = _EV.pair()
    if before is not None:
        before()
    kern.zo_perturb_layers(batch, 1e-3, events=lambda: pair)
    torch.cuda.synchronize()
    return _EV.elapsed_us(*pair)


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
    # what about the arena helps?  (i) the same arena with every tensor on a 2 MiB boundary (the
    # caching allocator's alignment, contiguous addresses), (ii) separate allocations whose starts
    # are staggered by odd multiples of 4 KiB + 256 B
    arena2 = torch.empty(total + 16 * (1 << 20), dtype=torch.float16, device="cuda")
    cur2 = [(-arena2.data_ptr() // 2) % (1 << 20)]

    def carve_2mb(shape):
        n = 1
        for d in shape:
            n *= d
        a = cur2[0]
        cur2[0] = (a + n + (1 << 20) - 1) // (1 << 20) * (1 << 20)
        return arena2[a:a + n].view(shape)

    batch_b, _ = build(kern, ws, carve_2mb)
    time_launch(kern, batch_b)
    out["one_arena_2MiB_aligned_tensors"] = stats([time_launch(kern, batch_b) for _ in range(20)], nbytes)
    keep, ctr = [], [0]

    def staggered(shape):
        n = 1
        for d in shape:
            n *= d
        ctr[0] += 1
        off = (2 * ctr[0] + 1) * 2048 + 128 * ctr[0]          # elements: odd multiples of 4 KiB + 256 B
        t = torch.empty(n + off + 64, dtype=torch.float16, device="cuda")
        keep.append(t)
        return t[off:off + n].view(shape)

    batch_c, _ = build(kern, ws, staggered)
    time_launch(kern, batch_c)
    out["separate_allocations_staggered_starts"] = stats([time_launch(kern, batch_c) for _ in range(20)], nbytes)
    ws_c = [staggered(tuple(w.shape)).copy_(w) for w in ws]
    batch_d, _ = build(kern, ws_c, staggered)
    time_launch(kern, batch_d)
    out["separate_staggered_incl_weights"] = stats([time_launch(kern, batch_d) for _ in range(20)], nbytes)
    # queued right behind MFMA load of different lengths (what the loop does: a block's K1 follows
    # the previous block's GEMM-heavy evaluations), and with a near-idle gap in between
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)

    def gemms(n):
        def go():
            for _ in range(n):
                a @ a
        return go

    def gemms_then_gap(n, cycles):
        def go():
            for _ in range(n):
                a @ a
            torch.cuda._sleep(cycles)
        return go

    for n in (10, 100):
        out[f"behind_{n}_gemms_8192"] = stats([time_launch(kern, batch_a, gemms(n)) for _ in range(8)], nbytes)
    out["behind_100_gemms_and_300us_idle"] = stats(
        [time_launch(kern, batch_a, gemms_then_gap(100, 600000)) for _ in range(8)], nbytes)
    out["behind_100_gemms_and_3ms_idle"] = stats(
        [time_launch(kern, batch_a, gemms_then_gap(100, 6000000)) for _ in range(8)], nbytes)
    torch.cuda.synchronize()
    out["idle_again"] = stats([time_launch(kern, batch_a) for _ in range(20)], nbytes)
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
