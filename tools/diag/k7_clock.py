#!/usr/bin/env python3
"""Where the time of one K7 matrix-mode call (a ViT-g block's four fp16 matrices, cold) goes:
phase stamps of workgroup 0 and the span of every kernel, from the library built by
tools/diag/k7_clock.sh (-DECO_K7_CLOCK).  Wall clock of the device, 100 MHz."""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ecoflap_amd import hip  # noqa: E402

hip.LIB_PATH = os.path.join(ROOT, "tools", "diag", "_build", "libecoflap_hip_k7clk.so")


def main():
    kern = hip.HipKernels()
    lib = hip.load_library()
    shapes = [(4224, 1408), (1408, 1408), (6144, 1408), (1408, 6144)]
    if "--one" in sys.argv:
        shapes = [(1408, 1408)]
    sets = 14
    wsets = [[(torch.randn(r, c, device="cuda") * 0.02).half() for r, c in shapes] for _ in range(sets)]
    srs = [torch.rand(c, device="cuda") + 0.05 for _, c in shapes]
    ks = [r * c // 2 for r, c in shapes]
    call = lambda i: kern.wanda_prune_block([(w, sr, "matrix", k, None) for w, sr, k in zip(wsets[i], srs, ks)])  # noqa: E731
    call(0); call(1)
    torch.cuda.synchronize()
    kern.wanda_fallback_counts()
    rows = []
    buf = (ctypes.c_ulonglong * 128)()
    for i in range(2, sets):
        assert lib.ecoflap_debug_k7_clock_reset() == 0
        torch.cuda.synchronize()
        torch.cuda._sleep(2000000)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); call(i); e.record()
        torch.cuda.synchronize()
        assert lib.ecoflap_debug_k7_clock_read(buf) == 0
        c = [[buf[k * 16 + j] for j in range(16)] for k in range(8)]
        rows.append((s.elapsed_time(e) * 1e3, c))
    print("fallbacks (bracket misses, crowded bins) over the timed calls:", kern.wanda_fallback_counts())
    med = lambda f: statistics.median(f(c) for _, c in rows)                      # noqa: E731
    us = lambda t: t / 100.0                                                       # noqa: E731
    print(f"events: median {statistics.median(t for t, _ in rows):.1f} us per call")
    t0 = lambda c: c[1][14]                                                        # noqa: E731
    names = [None, "sample + bracket", "apply2", "apply2 last workgroups"]
    for k, n in enumerate(names):
        if n is None:
            continue
        print(f"{n:24s} first entry {med(lambda c: us(c[k][14] - t0(c))):7.2f} us   last exit "
              f"{med(lambda c: us(c[k][15] - t0(c))):7.2f} us   span {med(lambda c: us(c[k][15] - c[k][14])):7.2f} us")
    print("sample + bracket, workgroup 0 (us since its entry; 1 loads issued + sqrt staged, 2 sample level 1, "
          "3 bracket known, 4 counted, 5 slot written):",
          " ".join(f"b{j}={med(lambda c: us(c[1][j] - c[1][0])):.2f}" for j in range(1, 6)))
    print("apply2, workgroup 0 (1 prologue done, 2 applied, 3 stores drained):", " ".join(f"a{j}={med(lambda c: us(c[2][j] - c[2][0])):.2f}" for j in range(1, 4)))


if __name__ == "__main__":
    main()
