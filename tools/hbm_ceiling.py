#!/usr/bin/env python3
"""Measured HBM stream ceilings of the box (the practical bound for K1, whose traffic is 32 parts
write to 1 part read): write-only (fill), copy (1 read : 1 write) and read-only (reduction) over a
713 MB working set (the size of one K1 launch on T5's largest matrix), cold between launches."""
import json
import statistics
import sys

import torch


def timed(fn, reps=7):
    blocker = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        for _ in range(4):
            blocker @ blocker
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) * 1e3)
    return statistics.median(out)


def main():
    n = 713 * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    b = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    spare = [torch.empty(n, dtype=torch.bfloat16, device="cuda") for _ in range(2)]  # evict MALL
    res = {}
    t = timed(lambda: a.fill_(1.0)); res["write_only_fill"] = 2 * n / t / 1e3
    t = timed(lambda: b.copy_(a)); res["copy_1r1w"] = 4 * n / t / 1e3
    t = timed(lambda: a.view(torch.int16).max()); res["read_only_max"] = 2 * n / t / 1e3
    t = timed(lambda: torch.add(a, b, out=spare[0])); res["add_2r1w"] = 6 * n / t / 1e3
    print(json.dumps({k: round(v) for k, v in res.items()}), "GB/s")


if __name__ == "__main__":
    main()
