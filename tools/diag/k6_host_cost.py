#!/usr/bin/env python3
"""K6 on a FlanT5 decoder block's 11 hooked inputs: where the time between two stream events goes.
Host cost of one `colsqnorm_accum_multi` call (argument marshalling), the stream time of eager calls
behind a busy stream, and the time of the same launch replayed from a HIP graph (how stage 2 runs
it: one capture per block, one replay per calibration sample).

    python3 tools/diag/k6_host_cost.py
"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ecoflap_amd import hip  # noqa: E402

BLOCKS = [("vit-g block: 4 inputs", torch.float16, [(8 * 257, 1408)] * 3 + [(8 * 257, 6144)]),
          ("t5 enc block: 7 inputs", torch.bfloat16, [(8 * 48, 2048)] * 6 + [(8 * 48, 5120)]),
          ("t5 dec block: 11 inputs", torch.bfloat16,
           [(8 * 16, 2048)] * 5 + [(8 * 48, 2048)] * 2 + [(8 * 16, 2048)] * 3 + [(8 * 16, 5120)])]


def main():
    kern = hip.HipKernels()
    for name, dt, shapes in BLOCKS:
        nbytes = sum(t * c * 2 for t, c in shapes)
        sets = 48
        xs = [[torch.randn(t, c, device="cuda").to(dt) for t, c in shapes] for _ in range(sets)]
        rows_ = [torch.zeros(c, device="cuda") for _, c in shapes]
        call = lambda i: kern.colsqnorm_accum_multi([(r, x, 8 * i, None, 8, False) for r, x in zip(rows_, xs[i])])  # noqa: E731
        for i in range(sets):
            call(i)
        torch.cuda.synchronize()
        # host cost per call (the device is far behind a long sleep: nothing blocks the enqueue)
        torch.cuda._sleep(400000000)
        t0 = time.perf_counter()
        for i in range(sets):
            call(i)
        host_us = (time.perf_counter() - t0) / sets * 1e6
        torch.cuda.synchronize()
        # stream time between events, eager, behind a busy stream
        out = []
        for _ in range(3):
            torch.cuda._sleep(400000000)
            evs = []
            for i in range(sets):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); call(i); e.record(); evs.append((s, e))
            torch.cuda.synchronize()
            out += [s.elapsed_time(e) * 1e3 for s, e in evs]
        # empty event pair: the floor of this way of timing
        torch.cuda._sleep(400000000)
        evs = []
        for i in range(sets):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); e.record(); evs.append((s, e))
        torch.cuda.synchronize()
        floor = statistics.median(s.elapsed_time(e) * 1e3 for s, e in evs)
        # the same launch from a HIP graph: 64 replays between one event pair
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            items = [(r, x, 0, None, 8, False) for r, x in zip(rows_, xs[0])]
            ws = kern.colsqnorm_multi_workspace(items)
            kern.colsqnorm_accum_multi(items, ws)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                kern.colsqnorm_accum_multi(items, ws)
            g.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(64):
                g.replay()
            e.record()
            torch.cuda.synchronize()
            graph_us = s.elapsed_time(e) * 1e3 / 64
        med, mn = statistics.median(out), min(out)
        print(f"{name:26s} {nbytes / 1e6:6.1f} MB  host {host_us:6.1f} us/call | eager stream median {med:6.1f} min {mn:6.1f} "
              f"(empty event pair {floor:4.1f}) | graph replay {graph_us:6.1f} us = {nbytes / graph_us / 8e6 * 100:4.1f} % of 8 TB/s",
              flush=True)
        del xs


if __name__ == "__main__":
    main()
