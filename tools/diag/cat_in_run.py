#!/usr/bin/env python3
"""Why does torch.cat take 0.5 ms inside the un-staged bench's worker threads (30 us alone)?  Wraps
torch.cat with a timer + the caching allocator's device-allocation counter and runs bench.py --unstaged."""
import collections
import runpy
import sys
import threading
import time

import torch

log = []
orig = torch.cat


def cat(tensors, *a, **k):
    n0 = torch.cuda.memory_stats().get("num_device_alloc", 0) if torch.cuda.is_initialized() else 0
    t0 = time.perf_counter()
    r = orig(tensors, *a, **k)
    dt = time.perf_counter() - t0
    n1 = torch.cuda.memory_stats().get("num_device_alloc", 0) if torch.cuda.is_initialized() else 0
    log.append((dt, tuple(tuple(x.shape) for x in tensors), str(r.dtype), n1 - n0, threading.current_thread().name))
    return r


torch.cat = cat
sys.argv = ["bench.py", "--gpus", "1", "--steps", "10", "--warmup", "3", "--unstaged", "--no-cpu-baseline", "--no-parity-leg"]
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
by = collections.defaultdict(list)
for dt, shapes, dtype, dn, th in log[len(log) // 3:]:
    by[(shapes, dtype)].append((dt, dn))
for key, v in sorted(by.items(), key=lambda kv: -sum(d for d, _ in kv[1]))[:8]:
    ds = sorted(d for d, _ in v)
    print(f"{len(v):6d} calls  median {ds[len(ds)//2]*1e6:8.1f} us  p90 {ds[int(len(ds)*.9)]*1e6:8.1f} us  total {sum(ds)*1e3:8.1f} ms  "
          f"device allocations {sum(n for _, n in v)}  {key}", file=sys.stderr)
