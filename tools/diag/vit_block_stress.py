#!/usr/bin/env python3
"""Is a captured fp16 EVA ViT-g block bit-reproducible replay after replay when TWO streams
replay their own copy of it at the same time (the two evaluation lanes)?  Counts, on the device,
the replays whose output differs from the first one; then again with the fused plumbing kernels
off, and per op class."""
import os, sys, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ecoflap_amd.shapes.eva_vit import Block, half_linear_weights
from ecoflap_amd.shapes import fused

def build(seed):
    torch.manual_seed(seed)
    blk = Block(1408, 16, 6144).eval().cuda()
    half_linear_weights(blk)
    return blk

def capture(blk, x, stream, chain):
    """graph of `chain` consecutive applications of the block (like the group chain)"""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream), torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for _ in range(2):
            y = x
            for _ in range(chain):
                y = blk(y)
        stream.synchronize()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16), \
            torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
        y = x
        for _ in range(chain):
            y = blk(y)
    return g, y

def run(batch, chain, iters, use_fused):
    if not use_fused:
        os.environ["ECOFLAP_NO_FUSED"] = "1"
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    blk_a, blk_b = build(0), build(0)
    xa = (torch.randn(batch, 257, 1408, device="cuda") * 0.5).half()
    xb = xa.clone()
    ga, ya = capture(blk_a, xa, sa, chain)
    gb, yb = capture(blk_b, xb, sb, chain)
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        ga.replay(); ref_a = ya.clone()
    with torch.cuda.stream(sb):
        gb.replay(); ref_b = yb.clone()
    torch.cuda.synchronize()
    same_streams = torch.equal(ref_a, ref_b)
    bad_a = torch.zeros((), dtype=torch.int64, device="cuda")
    bad_b = torch.zeros((), dtype=torch.int64, device="cuda")
    for _ in range(iters):
        with torch.cuda.stream(sa):
            ga.replay(); bad_a += (ya != ref_a).any()
        with torch.cuda.stream(sb):
            gb.replay(); bad_b += (yb != ref_b).any()
    torch.cuda.synchronize()
    print(f"batch {batch:3d} chain {chain} fused={use_fused}: lanes agree {same_streams}; "
          f"{int(bad_a)} + {int(bad_b)} of {iters} replays per lane differ from the lane's first")

def run_theta(batch, iters, which):
    """As the evaluation lanes do it: before every replay a different theta is copied into one
    weight matrix of the block (`which`), the output goes into a slot of a bigger buffer, and the
    slot is compared with what that theta gave the first time."""
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    lanes = []
    for st in (sa, sb):
        blk = build(0)
        x = (torch.randn(batch, 257, 1408, device="cuda", generator=torch.Generator("cuda").manual_seed(1)) * 0.5).half()
        w = dict(blk.named_parameters())[which].data
        thetas = [w.clone() + (i + 1) * 1e-3 * torch.randn_like(w) for i in range(4)]
        g, y = capture(blk, x, st, 1)
        slots = torch.zeros(4 * batch, 257, 1408, device="cuda", dtype=torch.float16)
        lanes.append((st, blk, w, thetas, g, y, slots))
    torch.cuda.synchronize()
    refs = []
    for st, blk, w, thetas, g, y, slots in lanes:
        r = []
        with torch.cuda.stream(st):
            for t in thetas:
                w.copy_(t); g.replay(); r.append(y.clone())
        refs.append(r)
    torch.cuda.synchronize()
    bad = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in lanes]
    for it in range(iters):
        for li, (st, blk, w, thetas, g, y, slots) in enumerate(lanes):
            with torch.cuda.stream(st):
                k = (it + li) % 4
                w.copy_(thetas[k])
                g.replay()
                slots[k * batch:(k + 1) * batch].copy_(y, non_blocking=True)
                bad[li] += (slots[k * batch:(k + 1) * batch] != refs[li][k]).any()
    torch.cuda.synchronize()
    print(f"theta into {which}: {int(bad[0])} + {int(bad[1])} of {iters} replays per lane differ")


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    run(8, 1, iters, True)
    run(32, 4, iters // 4, True)
    run(8, 4, iters // 2, True)
    for which in ("attn.qkv.weight", "mlp.fc2.weight"):
        run_theta(8, iters, which)
