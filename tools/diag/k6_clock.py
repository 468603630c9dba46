#!/usr/bin/env python3
"""Where the time of one K6 call on a ViT-g block's four hooked inputs (42.6 MB, one launch) goes:
the phase clock of tools/diag/k7_clock.sh (-DECO_K7_CLOCK), slots 5 (every workgroup up to its
ticket) and 6 (the column blocks' last workgroups)."""
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
    shapes = [(8 * 257, 1408)] * 3 + [(8 * 257, 6144)]
    sets = 14
    xs = [[torch.randn(t, c, device="cuda").half() for t, c in shapes] for _ in range(sets)]
    rows_ = [torch.zeros(c, device="cuda") for _, c in shapes]
    call = lambda i: kern.colsqnorm_accum_multi([(r, x, 8 * i, None, 8, False) for r, x in zip(rows_, xs[i])])  # noqa: E731
    call(0); call(1)
    torch.cuda.synchronize()
    out = []
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
        out.append((s.elapsed_time(e) * 1e3, c))
    med = lambda f: statistics.median(f(c) for _, c in out)                       # noqa: E731
    us = lambda t: t / 100.0                                                       # noqa: E731
    print(f"events: median {statistics.median(t for t, _ in out):.1f} us per call")
    print(f"all workgroups: first entry 0.00, last ticket taken {med(lambda c: us(c[5][15] - c[5][14])):.2f} us")
    print(f"last workgroups of the column blocks: first {med(lambda c: us(c[6][14] - c[5][14])):.2f} us, "
          f"last done {med(lambda c: us(c[6][15] - c[5][14])):.2f} us")
    print("workgroup 0 (us since its entry; 1 rows streamed, 2 partials stored + drained, 3 ticket back):",
          " ".join(f"p{j}={med(lambda c: us(c[5][j] - c[5][0])):.2f}" for j in range(1, 4)))


if __name__ == "__main__":
    main()
