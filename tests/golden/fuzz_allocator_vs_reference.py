#!/usr/bin/env python3
"""One-off check run in the build container (the reference cannot travel): the 400 random allocation
problems of tests/test_cabi_exports.py::test_allocator_through_cabi_equals_the_oracle_on_random_problems
through the REFERENCE's `LayerSparsity.compute_the_sparsity_per_group`
(UPop/pruners/layer_single_base_pruner.py:247-314, imported from /root/reference), the oracle
restatement and the C++ allocator behind the C ABI: all three equal, float bit for float bit.
    python3 tests/golden/fuzz_allocator_vs_reference.py"""
import os
import signal
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import allocator as oracle_alloc  # noqa: E402
import make_golden  # noqa: E402
from ecoflap_amd import hip  # noqa: E402


class _Timeout(Exception):
    pass


def _alarm(*a):
    raise _Timeout()


def main():
    LayerSparsity, _ = make_golden.import_upop_pruners()
    rng = np.random.default_rng(20261004)
    same = skipped = bad = all_nan = 0
    for case in range(400):
        G = int(rng.integers(1, 400)) if case % 7 else int(rng.integers(1, 4))
        nums = (10 ** rng.uniform(2, 8, size=G)).astype(np.int64)
        kind = case % 6
        if kind == 0:
            scores = rng.random(G)
        elif kind == 1:
            scores = np.round(rng.random(G) * 4) / 4
        elif kind == 2:
            scores = 10 ** rng.uniform(-12, 12, size=G)
        elif kind == 3:
            scores = rng.random(G) * (rng.random(G) < 0.3)
        elif kind == 4:
            scores = np.full(G, 1.0)
        else:
            scores = rng.random(G)
            scores[rng.integers(0, G)] = np.nan
        scores = scores.astype(np.float32)
        mx = float(rng.choice([0.5, 0.6, 0.8, 0.9, 1.0]))
        target = float(rng.uniform(0.1, min(0.95, mx)))
        keep = int(nums.sum() * (1 - target))
        try:
            want, _ = oracle_alloc.compute_sparsity_per_group(keep, scores, nums, mx, max_iters=2000)
        except RuntimeError:
            skipped += 1
            continue
        gs = {f"g{i}": torch.tensor(float(s), dtype=torch.float32) for i, s in enumerate(scores)}
        gn = {f"g{i}": int(n) for i, n in enumerate(nums)}
        signal.signal(signal.SIGALRM, _alarm)
        signal.alarm(10)
        try:
            res = LayerSparsity.compute_the_sparsity_per_group(None, keep, gs, gn, mx)
        except _Timeout:
            print("reference does not terminate on case", case)
            skipped += 1
            continue
        finally:
            signal.alarm(0)
        ref = np.array([float(res[f"g{i}"]) for i in range(G)], dtype=np.float32)
        got, _ = hip.allocate_sparsity(scores, nums, keep, mx)
        fa, fb, fc = (np.array(x, dtype=np.float32) for x in (ref, want, got))
        nan = np.isnan(fa)
        # (a NaN is a NaN: the reference's 0/0 and numpy's differ in the sign bit only)
        ok = (np.array_equal(nan, np.isnan(fb)) and np.array_equal(nan, np.isnan(fc))
              and np.array_equal(fa[~nan].view(np.uint32), fb[~nan].view(np.uint32))
              and np.array_equal(fa[~nan].view(np.uint32), fc[~nan].view(np.uint32)))
        if not ok:
            print("DIFFERENT: case", case, "kind", kind, "G", G, "mx", mx, "target", target)
            bad += 1
            continue
        all_nan += int(nan.all())
        same += 1
    print(f"{bad} DIFFERENT; {same} problems: reference == oracle == C ABI bit for bit ({all_nan} of them all-NaN tables: "
          f"NaN or all-zero scores); {skipped} skipped (non-terminating for the oracle / reference)")


if __name__ == "__main__":
    main()
