#!/usr/bin/env python3
"""ViT-g attention at the loop's batch sizes: this build's kernel vs torch's fused attention
(HIP events, 30 launches each)."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
from ecoflap_amd.shapes import fused  # noqa: E402


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    H, D, N = 16, 88, 257
    rows = []
    for B in (8, 32, 128):
        qkv = (torch.randn(B, N, 3 * H * D, device="cuda") * 0.7).half()
        flops = 4.0 * B * H * N * N * D
        with torch.no_grad():
            t_own = timed(lambda: fused.vit_attention(qkv, H, D ** -0.5))

            def lib():
                q, k, v = qkv.reshape(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
                return F.scaled_dot_product_attention(q, k, v, scale=D ** -0.5).transpose(1, 2).reshape(B, N, -1)
            t_lib = timed(lib)
        rows.append({"batch": B, "own_us": round(t_own, 1), "library_us": round(t_lib, 1),
                     "own_tflops": round(flops / t_own / 1e6, 1), "library_tflops": round(flops / t_lib / 1e6, 1),
                     "hbm_bytes": 4 * B * N * H * D * 2,
                     "own_pct_of_8TBs": round(100 * 4 * B * N * H * D * 2 / (t_own * 1e-6) / 8e12, 1)})
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
