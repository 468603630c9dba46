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

