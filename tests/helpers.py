import numpy as np
import torch

NP2T = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def from_bits(a, dtype):
    """int view saved by tests/golden/make_golden.py -> tensor of `dtype`."""
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype == torch.float32:
        return t.view(torch.float32)
    if dtype in (torch.float16, torch.bfloat16):
        return t.view(dtype)
    return t


def to_bits(t):
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.float32:
        return t.view(torch.int32).numpy()
    if t.dtype in (torch.float16, torch.bfloat16):
        return t.view(torch.int16).numpy()
    return t.numpy()


def assert_same_pruning(a, b, rel=1e-6, slack=2):
    """Pruned state_dicts `a` (HIP) and `b` (oracle) after a Wanda run on the same GPU forward.

    The column statistic is a float reduction: the oracle adds the squares in torch's CPU order
    (one fp32 fma chain per column), a GPU kernel necessarily in another, and the two agree to a
    few ulps, not bit for bit.  The selection then compares |W| * sqrt(stat) — at true row
    lengths, where bf16 / fp16 weights take few distinct magnitudes, a row's k-th and (k+1)-th
    smallest metrics are now and then within that ulp, and which of the two is pruned flips.
    So: every 2-D tensor has the SAME NUMBER of zeros, and the positions differ in at most
    `slack + rel * numel` elements per tensor (a swap is two positions).  -> number of tensors
    that were not bit-identical, total differing positions."""
    import torch
    differing, positions = 0, 0
    assert a.keys() == b.keys()
    for k in a:
        x, y = a[k], b[k]
        if x.dim() != 2 or torch.equal(x, y):
            assert torch.equal(x, y), k
            continue
        assert int((x == 0).sum()) == int((y == 0).sum()), f"{k}: zero counts differ"
        d = int((x != y).sum())
        assert d <= slack + rel * x.numel(), f"{k}: {d} of {x.numel()} elements differ"
        differing += 1
        positions += d
    return differing, positions
