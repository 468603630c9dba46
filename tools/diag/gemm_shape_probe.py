#!/usr/bin/env python3
"""Per GEMM shape of the BLIP-2 loop: which library kernel runs, how long it takes, a checksum
of its output (same across library settings <=> same arithmetic order), and whether the rows of
ONE evaluation come out bit-identical when 4 / 16 evaluations are concatenated (batch
invariance).  Run once per library setting (TENSILE_STREAMK_* in the environment)."""
import hashlib
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from streamk_gemm_stress import kernel_names  # noqa: E402

SHAPES = [  # (name, rows of one evaluation, K, N, dtype, bias)
    ("vit.qkv", 2056, 1408, 4224, torch.float16, False),
    ("vit.proj", 2056, 1408, 1408, torch.float16, True),
    ("vit.fc1", 2056, 1408, 6144, torch.float16, True),
    ("vit.fc2", 2056, 6144, 1408, torch.float16, True),
    ("t5.enc.qkvo", 384, 2048, 2048, torch.bfloat16, False),
    ("t5.enc.wi", 384, 2048, 5120, torch.bfloat16, False),
    ("t5.enc.wo", 384, 5120, 2048, torch.bfloat16, False),
    ("t5.dec.qkvo", 128, 2048, 2048, torch.bfloat16, False),
    ("t5.lm_head", 128, 2048, 32128, torch.bfloat16, False),
]


def timed(fn, reps=30):
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
    env = {k: v for k, v in os.environ.items() if k.startswith(("TENSILE", "HIPBLASLT", "TORCH_BLAS"))}
    rows = []
    for name, m1, K, N, dt, bias in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(7)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(dt)
        b = (torch.randn(N, device="cuda", generator=g) * 0.02).to(dt) if bias else None
        x16 = (torch.randn(16 * m1, K, device="cuda", generator=g) * 0.7).to(dt)
        rec = {"shape": name, "K": K, "N": N}
        alone = [F.linear(x16[i * m1:(i + 1) * m1].contiguous(), w, b) for i in (0, 3, 15)]
        for k in (1, 4, 16):
            x = x16[:k * m1].contiguous()
            y = F.linear(x, w, b)
            rec[f"k{k}_us"] = round(timed(lambda: F.linear(x, w, b)), 1)
            rec[f"k{k}_kernel"] = ";".join(sorted(n[:70] + ("..SK" if "_SK" in n else "") for n in kernel_names(lambda: F.linear(x, w, b))))
            rec[f"k{k}_sha"] = hashlib.sha256(y.cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]
            if k > 1:
                ok = torch.equal(y[:m1], alone[0]) and (k <= 3 or torch.equal(y[3 * m1:4 * m1], alone[1])) \
                    and (k < 16 or torch.equal(y[15 * m1:], alone[2]))
                rec[f"k{k}_invariant"] = bool(ok)
        rows.append(rec)
        del w, x16
    print("PROBE " + json.dumps({"env": env, "shapes": rows}), flush=True)


if __name__ == "__main__":
    main()
