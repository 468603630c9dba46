#!/usr/bin/env python3
"""Reproducer for the transient loss mismatch of round 2 (root cause, round 3).

tools/diag/transient_hunt.py --subops localised every event (12 of 12) to ONE op: the output
of the EVA ViT-g block's fc1 Linear at 4 concatenated evaluations (fp16 GEMM + bias,
M = 32*257 = 8224, N = 6144, K = 1408), which torch sends to hipBLASLt's Stream-K kernel
`Custom_Cijk_Alik_Bljk_HHS_BH_Bias_HA_S_SAV_NTD_SK3_UserArgs_MT256x256x64_MI16x16x1`.  The
differing elements are always a fragment of one 256x256 macro tile — 8-row slivers at a stride
of 32 rows inside one 128-row half, <= 252 columns wide — i.e. part of a partial tile that one
workgroup hands to another through the Stream-K workspace arrived stale.

This script runs that GEMM alone: `--streams` streams each replay a captured graph of
`F.linear` calls on rotating inputs and count, on the device, the calls whose output differs
from the first result for that input; it prints the rate and the row / column pattern of the
differing elements.  Environment knobs of the library (TENSILE_STREAMK_*) are passed through,
so the same command shows which setting makes the kernel safe.

    python3 tools/diag/streamk_gemm_stress.py --iters 200000 --streams 2
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F


def kernel_names(fn):
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            fn()
            torch.cuda.synchronize()
        names = {}
        for e in prof.events():
            if e.device_type is not None and str(e.device_type).endswith("CUDA") and "Cijk" in e.name:
                names[e.name[:140]] = names.get(e.name[:140], 0) + 1
        return names
    except Exception as e:       # noqa: BLE001
        return {"profiler_error": repr(e)}


def ranges(ids):
    out = []
    for v in ids:
        if out and v == out[-1][1] + 1:
            out[-1][1] = v
        else:
            out.append([v, v])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=8224)
    ap.add_argument("--N", type=int, default=6144)
    ap.add_argument("--K", type=int, default=1408)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--iters", type=int, default=100000, help="GEMM calls per stream")
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--per-graph", type=int, default=16)
    ap.add_argument("--inputs", type=int, default=4)
    ap.add_argument("--no-bias", action="store_true")
    ap.add_argument("--tag", default="")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    dev = "cuda"
    env = {k: v for k, v in os.environ.items()
           if k.startswith(("TENSILE", "HIPBLASLT", "ROCBLAS", "TORCH_BLAS", "PYTORCH_TUNABLEOP"))}
    lanes = []
    for s in range(args.streams):
        g = torch.Generator(device=dev).manual_seed(100 + s)
        w = (torch.randn(args.N, args.K, device=dev, generator=g) * 0.02).to(dt)
        b = None if args.no_bias else (torch.randn(args.N, device=dev, generator=g) * 0.02).to(dt)
        xs = [(torch.randn(args.M, args.K, device=dev, generator=g) * 0.7).to(dt)
              for _ in range(args.inputs)]
        lanes.append({"stream": torch.cuda.Stream(), "w": w, "b": b, "xs": xs})
    torch.cuda.synchronize()
    names = kernel_names(lambda: F.linear(lanes[0]["xs"][0], lanes[0]["w"], lanes[0]["b"]))
    # references: the first result per input, confirmed three times with the device otherwise idle
    for ln in lanes:
        ln["refs"] = []
        for x in ln["xs"]:
            r = F.linear(x, ln["w"], ln["b"])
            for _ in range(3):
                assert torch.equal(r, F.linear(x, ln["w"], ln["b"])), "not reproducible when idle"
            ln["refs"].append(r)
        ln["bad"] = torch.zeros((), dtype=torch.int64, device=dev)
        ln["mask"] = torch.zeros(args.M, args.N, dtype=torch.bool, device=dev)
    torch.cuda.synchronize()
    # one graph per lane: per_graph calls, each compared with its reference on the device
    for ln in lanes:
        st = ln["stream"]
        with torch.cuda.stream(st):
            for i in range(2):
                F.linear(ln["xs"][i % args.inputs], ln["w"], ln["b"])
        st.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st, capture_error_mode="thread_local"):
            for i in range(args.per_graph):
                y = F.linear(ln["xs"][i % args.inputs], ln["w"], ln["b"])
                d = y != ln["refs"][i % args.inputs]
                ln["bad"] += d.any()
                ln["mask"] |= d
        ln["graph"] = graph
    torch.cuda.synchronize()
    # timing of one call (idle device, eager)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        F.linear(lanes[0]["xs"][0], lanes[0]["w"], lanes[0]["b"])
    e1.record()
    torch.cuda.synchronize()
    us_per_call = e0.elapsed_time(e1) * 1e3 / 50
    # batch invariance: rows of the first eighth alone == the same rows inside the full problem
    m8 = args.M // 4
    inv = {}
    for frac, m in (("quarter", args.M // 4), ("half", args.M // 2)):
        a = F.linear(lanes[0]["xs"][0][:m].contiguous(), lanes[0]["w"], lanes[0]["b"])
        inv[frac] = bool(torch.equal(a, lanes[0]["refs"][0][:m]))
    t0 = time.time()
    n_replays = (args.iters + args.per_graph - 1) // args.per_graph
    for r in range(n_replays):
        for ln in lanes:
            with torch.cuda.stream(ln["stream"]):
                ln["graph"].replay()
        if r % 256 == 255:
            torch.cuda.synchronize()          # bound the queue
    torch.cuda.synchronize()
    secs = time.time() - t0
    out = {"tag": args.tag, "M": args.M, "N": args.N, "K": args.K, "dtype": args.dtype,
           "streams": args.streams, "calls_per_stream": n_replays * args.per_graph,
           "seconds": round(secs, 1), "us_per_call_idle": round(us_per_call, 1),
           "kernels": names, "env": env, "batch_invariant_rows": inv,
           "bad_calls": [int(ln["bad"]) for ln in lanes], "patterns": []}
    for ln in lanes:
        if int(ln["bad"]):
            idx = ln["mask"].nonzero()
            rows = idx[:, 0].unique().tolist()
            cols = idx[:, 1].unique().tolist()
            out["patterns"].append({"n_elements": int(idx.shape[0]), "rows": ranges(rows)[:40],
                                    "cols": ranges(cols)[:40]})
    print("STRESS " + json.dumps(out), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "a") as f:
            f.write(json.dumps(out) + "\n")


if __name__ == "__main__":
    main()
