#!/usr/bin/env python3
"""Time the Wanda kernels (K6 column statistic, K7 rows / matrix selection) on BLIP-2's
shapes with cold inputs (rotating buffer sets > 256 MiB) and print GB/s against the
algorithmic bytes of DESIGN.md §4.  Also the workload for rocprofv3 summaries."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecoflap_amd import hip  # noqa: E402


def timed(fn, n_sets, reps=3, fresh=False):
    """fresh: the call changes its input (pruning): warm up on set 0 only, time sets 1..n-1 once."""
    for i in range(1 if fresh else n_sets):
        fn(i)
    torch.cuda.synchronize()
    out = []
    for _ in range(1 if fresh else reps):
        # keeps the stream busy while the host enqueues (no GEMMs: they would pull the clock down
        # for what follows); ~100 us per call, or a cheap call with many arguments (the FlanT5
        # decoder block's 11 K6 inputs: 7 us of kernel, 30 us of Python) reads as the host's time
        torch.cuda._sleep(max(2000000, 240000 * n_sets))
        evs = []
        for i in range(1 if fresh else 0, n_sets):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(i); e.record(); evs.append((s, e))
        torch.cuda.synchronize()
        out += [s.elapsed_time(e) * 1e3 for s, e in evs]
    return statistics.median(out), min(out)


def main():
    # --only k6,rows,matrix,syrk,matrixblock,rowsblock: sections to run (default all)
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    want = lambda tag: only is None or tag in only          # noqa: E731
    kern = hip.HipKernels()
    res = []
    # K6: hooked Linear inputs of one calibration batch (bs 8)
    for name, tokens, cols, dt in [("vit qkv/fc1 input", 8 * 257, 1408, torch.float32),
                                   ("vit fc2 input", 8 * 257, 6144, torch.float16),
                                   ("t5 enc q/k/v/wi input", 8 * 48, 2048, torch.bfloat16),
                                   ("t5 enc wo input", 8 * 48, 5120, torch.bfloat16)]:
        if not want("k6"):
            break
        nbytes = tokens * cols * torch.empty(0, dtype=dt).element_size()
        sets = max(2, int(6e8 // nbytes))
        xs = [torch.randn(tokens, cols, device="cuda").to(dt) for _ in range(sets)]
        s = torch.zeros(cols, device="cuda")
        med, mn = timed(lambda i: kern.colsqnorm_accum(s, xs[i], 8 * i, 8), sets)
        res.append(("K6 colsqnorm", name, nbytes, med, mn))
    # K6, ONE launch for all hooked inputs of a transformer block (one calibration batch, bs 8)
    for name, dt, shapes in [("vit-g block: 4 inputs", torch.float16,
                              [(8 * 257, 1408)] * 3 + [(8 * 257, 6144)]),
                             ("t5 enc block: 7 inputs", torch.bfloat16,
                              [(8 * 48, 2048)] * 6 + [(8 * 48, 5120)]),
                             ("t5 dec block: 11 inputs", torch.bfloat16,
                              [(8 * 16, 2048)] * 4 + [(8 * 16, 2048)] + [(8 * 48, 2048)] * 2
                              + [(8 * 16, 2048)] * 3 + [(8 * 16, 5120)])]:
        if not want("k6"):
            break
        nbytes = sum(t * c * 2 for t, c in shapes)
        sets = max(2, int(6e8 // nbytes))
        xs = [[torch.randn(t, c, device="cuda").to(dt) for t, c in shapes] for _ in range(sets)]
        rows_ = [torch.zeros(c, device="cuda") for _, c in shapes]
        fn = lambda i: kern.colsqnorm_accum_multi(                               # noqa: E731
            [(r, x, 8 * i, None, 8, False) for r, x in zip(rows_, xs[i])])
        med, mn = timed(fn, sets)
        res.append(("K6 multi", name, nbytes, med, mn))
        del xs
    # K7 rows: T5 matrices (bf16); matrix mode: ViT matrices (fp16)
    for name, rows, cols, dt, mode in [("t5 wo 2048x5120", 2048, 5120, torch.bfloat16, "rows"),
                                       ("t5 wi 5120x2048", 5120, 2048, torch.bfloat16, "rows"),
                                       ("t5 q 2048x2048", 2048, 2048, torch.bfloat16, "rows"),
                                       ("vit fc1 6144x1408", 6144, 1408, torch.float16, "matrix"),
                                       ("vit qkv 4224x1408", 4224, 1408, torch.float16, "matrix"),
                                       ("vit proj 1408x1408", 1408, 1408, torch.float16, "matrix")]:
        if not want(mode):
            continue
        es = torch.empty(0, dtype=dt).element_size()
        nbytes = 2 * es * rows * cols + 4 * cols
        sets = max(3, int(6e8 // (rows * cols * es)))
        ws = [(torch.randn(rows, cols, device="cuda") * 0.02).to(dt) for _ in range(sets)]
        sr = torch.rand(cols, device="cuda") + 0.05
        if mode == "rows":
            fn = lambda i: kern.wanda_prune_rows(ws[i], sr, cols // 2)          # noqa: E731
        else:
            fn = lambda i: kern.wanda_prune_matrix(ws[i], sr, rows * cols // 2)  # noqa: E731
        med, mn = timed(fn, sets, fresh=True)
        res.append((f"K7 {mode}", name, nbytes, med, mn))
        del ws
    # K7 per transformer block: all its Linears through ecoflap_wanda_prune_block
    blocks = [("vit-g block (4 fp16 matrices)", "matrix", torch.float16,
               [(4224, 1408), (1408, 1408), (6144, 1408), (1408, 6144)]),
              ("t5 encoder block (7 bf16)", "rows", torch.bfloat16,
               [(2048, 2048)] * 4 + [(5120, 2048)] * 2 + [(2048, 5120)]),
              ("t5 decoder block (11 bf16)", "rows", torch.bfloat16,
               [(2048, 2048)] * 8 + [(5120, 2048)] * 2 + [(2048, 5120)])]
    for name, mode, dt, shapes in blocks:
        if not (want(mode) or want(mode + "block")):      # --only matrixblock: the block call alone
            continue
        es = 2
        nbytes = sum(2 * es * r * c + 4 * c for r, c in shapes)
        sets = max(3, int(8e8 // (nbytes // 2)))
        wsets = [[(torch.randn(r, c, device="cuda") * 0.02).to(dt) for r, c in shapes]
                 for _ in range(sets)]
        srs = [torch.rand(c, device="cuda") + 0.05 for _, c in shapes]
        ks = [int((c if mode == "rows" else r * c) * 0.5) for r, c in shapes]
        fn = lambda i: kern.wanda_prune_block(                                   # noqa: E731
            [(w, sr, mode, k, None) for w, sr, k in zip(wsets[i], srs, ks)])
        if mode == "matrix":
            kern.wanda_fallback_counts()
        med, mn = timed(fn, sets, fresh=True)
        if mode == "matrix":
            name += "  [fallbacks (misses, crowded): %d, %d]" % kern.wanda_fallback_counts()
        res.append((f"K7 {mode} block", name, nbytes, med, mn))
        if mode == "matrix":
            # the three-histogram selection (4 reads + 1 write), for comparison
            os.environ["ECOFLAP_WANDA_SAMPLED"] = "0"
            wsets2 = [[(torch.randn(r, c, device="cuda") * 0.02).to(dt) for r, c in shapes]
                      for _ in range(sets)]
            fn2 = lambda i: kern.wanda_prune_block(                              # noqa: E731
                [(w, sr, mode, k, None) for w, sr, k in zip(wsets2[i], srs, ks)])
            med, mn = timed(fn2, sets, fresh=True)
            del os.environ["ECOFLAP_WANDA_SAMPLED"]
            res.append((f"K7 {mode} block", name + " ECOFLAP_WANDA_SAMPLED=0", nbytes, med, mn))
            del wsets2
        del wsets
    # SparseGPT Hessian (MFMA SYRK): flops against the dense fp16 / bf16 peak
    flops_rows = []
    for name, tokens, cols, dt in [("vit fc2 input [2056, 6144] fp16", 8 * 257, 6144, torch.float16),
                                   ("vit qkv input [2056, 1408] fp16", 8 * 257, 1408, torch.float16),
                                   ("t5 wo input [384, 5120] bf16", 384, 5120, torch.bfloat16)]:
        if not want("syrk"):
            break
        xs = [torch.randn(tokens, cols, device="cuda").to(dt) for _ in range(3)]
        H = torch.zeros(cols, cols, device="cuda")
        med, mn = timed(lambda i: kern.hessian_accum(H, xs[i], 8 * i, 8), 3)
        nt = -(-cols // 128)
        flops = 2.0 * tokens * 128 * 128 * nt * (nt + 1) / 2        # tiles actually computed
        flops_rows.append((name, flops, med))
        # eight calibration samples per call (SparseGPT.samples_per_call): H read / written once
        x8 = torch.cat([xs[i % 3] for i in range(8)], 0)
        med8, _ = timed(lambda i: kern.hessian_accum(H, x8, 64 * i, 64), 3)
        flops_rows.append((name + " x8 samples per call", 8 * flops, med8))
        import math
        xf = [x.float() for x in xs]
        def ref(i):
            H.mul_(0.5)
            H.addmm_(xf[i].t(), xf[i])
        medr, _ = timed(ref, 3)
        flops_rows.append((name + " (torch fp32 addmm, full square)", 2.0 * tokens * cols * cols, medr))
    for name, flops, med in flops_rows:
        print(f"{'Hessian SYRK':14s} {name:52s} {flops/1e9:8.1f} GFLOP  median {med:8.1f} us  "
              f"{flops/med/1e6:7.1f} TFLOP/s ({flops/med/1e6/2500*100:5.1f}% of 2.5 PFLOP/s dense)")
    for k, name, nbytes, med, mn in res:
        print(f"{k:14s} {name:24s} {nbytes/1e6:8.1f} MB  median {med:8.1f} us  min {mn:8.1f} us  "
              f"{nbytes/med/1e3:7.0f} GB/s ({nbytes/med/1e3/80:5.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
