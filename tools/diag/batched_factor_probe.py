#!/usr/bin/env python3
"""Would ONE batched torch.linalg call over a block's same-size Hessians (a single solver call: safe)
give the one-matrix results bit for bit, and what does it cost?  n = 1408 x 3 (a ViT-g block's
qkv / proj / fc1 inputs) and n = 2048 x 6 (a FlanT5 decoder block)."""
import statistics
import time

import torch


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(1)
    for n, b in ((1408, 3), (2048, 6), (5120, 2)):
        hs = []
        for i in range(b):
            x = torch.randn(4 * n, n, device="cuda", generator=g)
            h = x.t() @ x / (4 * n)
            h += 0.01 * torch.mean(torch.diag(h)) * torch.eye(n, device="cuda")
            hs.append(h)
        stack = torch.stack(hs)
        one = [torch.linalg.cholesky_ex(h)[0] for h in hs]
        bat = torch.linalg.cholesky_ex(stack)[0]
        same_l = all(torch.equal(a, bat[i]) for i, a in enumerate(one))
        inv1 = [torch.cholesky_inverse(l) for l in one]
        invb = torch.cholesky_inverse(torch.stack(one))
        same_i = all(torch.equal(a, invb[i]) for i, a in enumerate(inv1))
        up1 = [torch.linalg.cholesky_ex(a, upper=True)[0] for a in inv1]
        upb = torch.linalg.cholesky_ex(torch.stack(inv1), upper=True)[0]
        same_u = all(torch.equal(a, upb[i]) for i, a in enumerate(up1))
        t1 = timed(lambda: [torch.linalg.cholesky_ex(torch.cholesky_inverse(torch.linalg.cholesky_ex(h)[0]), upper=True) for h in hs])
        tb = timed(lambda: torch.linalg.cholesky_ex(torch.cholesky_inverse(torch.linalg.cholesky_ex(stack)[0]), upper=True))
        print(f"n {n} x {b}: batched == one by one: lower {same_l}, inverse {same_i}, upper {same_u}; chain one by one {t1:.1f} ms, batched {tb:.1f} ms")


if __name__ == "__main__":
    main()
