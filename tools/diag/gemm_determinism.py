#!/usr/bin/env python3
"""Are the library's fp16 / bf16 / fp32 GEMMs of the loop bit-reproducible call to call — alone
and with another stream keeping the device busy (the second evaluation lane)?"""
import torch
torch.manual_seed(0)
dev = "cuda"
cases = [("vit qkv  fp16 M=8224", 8224, 1408, 4224, torch.float16),
         ("vit fc1  fp16 M=8224", 8224, 1408, 6144, torch.float16),
         ("vit fc2  fp16 M=8224", 8224, 6144, 1408, torch.float16),
         ("vit qkv  fp16 M=2056", 2056, 1408, 4224, torch.float16),
         ("t5 wi    bf16 M=6144", 6144, 2048, 5120, torch.bfloat16),
         ("t5 q     bf16 M=384", 384, 2048, 2048, torch.bfloat16),
         ("qformer kv fp32 M=2056", 2056, 1408, 768, torch.float32)]
side = torch.cuda.Stream()
noise_a = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
for name, M, K, N, dt in cases:
    x = torch.randn(M, K, device=dev).to(dt)
    w = torch.randn(N, K, device=dev).to(dt) * 0.02
    b = torch.randn(N, device=dev).to(dt)
    ref = torch.nn.functional.linear(x, w, b)
    torch.cuda.synchronize()
    for label, busy in (("alone", False), ("with a busy second stream", True)):
        diff = 0
        for it in range(300):
            if busy:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        noise_a @ noise_a
            out = torch.nn.functional.linear(x, w, b)
            if not torch.equal(out, ref):
                diff += 1
        torch.cuda.synchronize()
        print(f"{name:26s} {label:28s} {diff:3d} of 300 calls differ from the first")
