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


def free_port():
    """A TCP port nobody holds right now, from the kernel (bind to 0).  The rendezvous ports of the
    multi-process tests used to be computed from the pid inside 33500-42500, i.e. inside Linux's
    ephemeral range: any outgoing connection of any process on the box — the previous test's gloo
    pairs in TIME_WAIT included — could be sitting on one, and the run then failed in the
    TCPStore's bind."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
