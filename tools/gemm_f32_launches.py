#!/usr/bin/env python3
"""Time the fp32 MFMA GEMM of csrc/gemm_f32.hip against torch's (hipBLASLt) on the Q-Former's
shapes at 16 evaluations (batch size 8 and 1); TFLOP/s against the 157 TFLOP/s fp32 MFMA peak."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ECOFLAP_PINNED_GEMM"] = "1"
from ecoflap_amd.shapes import fused  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def timed(fn, reps=15, per=10):
    """Median over `reps` of the mean of `per` back-to-back launches (one pair of events around
    them: a single launch's figure would carry the launch gap), microseconds."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(per):
            fn()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) * 1e3 / per)
    return statistics.median(out)


for name, M, N, K in [("self-attn q/k/v/o, 16 x 8 x 32 rows", 4096, 768, 768),
                      ("intermediate, 16 x 8 x 32 rows", 4096, 3072, 768),
                      ("output, 16 x 8 x 32 rows", 4096, 768, 3072),
                      ("cross-attn k/v, 16 x 8 x 257 rows", 32896, 768, 1408),
                      ("t5_proj, 16 x 8 x 32 rows", 4096, 2048, 768),
                      ("self-attn q/k/v/o, 16 x 1 x 32 rows", 512, 768, 768),
                      ("cross-attn k/v, 16 x 1 x 257 rows", 4112, 768, 1408),
                      ("intermediate, one evaluation at batch 1", 32, 3072, 768)]:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    fl = 2.0 * M * N * K
    with torch.no_grad():
        ref = F.linear(x.double(), w.double(), b.double())
        t_lib = timed(lambda: F.linear(x, w, b))
        lib_err = ((F.linear(x, w, b).double() - ref).abs().max() / ref.abs().max()).item()
        y = fused.linear(x, w, b)
        err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
        t_own = timed(lambda: fused.linear(x, w, b))
        print(f"{name:42s} [{M:6d} x {K:4d}] x {N:4d} own {t_own:7.1f} us = "
              f"{fl / t_own / 1e6:6.1f} TFLOP/s ({fl / t_own / 1e6 / 157 * 100:4.1f} % of 157) err {err:.1e}   "
              f"library {t_lib:7.1f} us = {fl / t_lib / 1e6:6.1f} TFLOP/s err {lib_err:.1e}", flush=True)
