"""The GEMM library's Stream-K kernels and this loop (root cause of round 2's transient).

torch sends the loop's larger fp16 / bf16 Linears to hipBLASLt's Stream-K kernels
(`Custom_Cijk_..._SK3_UserArgs_MT256x256x64_MI16x16x1`: the fc1 GEMM of an EVA ViT-g block from
one evaluation upwards, qkv / proj / fc2 from four, FlanT5's wi and the lm_head at sixteen).
In their default mode those kernels

  * are NOT bit-reproducible call to call: about one call in 30 000 of the ViT-g fc1 shape
    (M = 8224, N = 6144, K = 1408, fp16 + bias) returns a fragment of one 256x256 macro tile —
    8-row slivers at a stride of 32 rows, up to 252 columns — that differs from every other call
    on the same inputs; one stream suffices, two concurrent streams raise the rate
    (`tools/diag/streamk_gemm_stress.py`: 8 of 120 000 calls on two streams, 1 of 60 000 on one;
    `tools/diag/transient_hunt.py --subops`: all 12 loss mismatches of 20 full ViT passes start at
    that op).  That is the "one slot of one chunk off by ~7e-3" of round 2, and it can hit any
    forward of the run (stage 2, first order, the sequential path) just as well;
  * are not batch invariant at 16 concatenated evaluations (the K split of a tile depends on
    its position in the grid), which kept the ViT-g blocks from being shared.

With `TENSILE_STREAMK_DATA_PARALLEL=1` the same kernels give every workgroup whole tiles: 0
differing calls in 800 000 on two streams, bit-identical results at 1 and 4 evaluations, batch
invariant at 16, and not slower (fc1 at 16 evaluations: 516 vs 536 us; `profiles/r03_streamk/`).
The library reads the variable when it is first used, so it has to be in the environment before
the process's first GEMM: `configure()` runs at package import, and `verify()` — called by the
loss closures before the first captured forward on the GPU — checks that the setting is LIVE
(a 16-evaluation qkv-shaped GEMM is batch invariant only in that mode) and refuses to run
otherwise.  `ECOFLAP_ALLOW_STREAMK=1` turns the refusal into a warning."""
import os
import warnings

ENV = "TENSILE_STREAMK_DATA_PARALLEL"
_verified = {}


def configure():
    """Put the setting into the environment unless the user already chose one."""
    os.environ.setdefault(ENV, "1")


def verify(device=None):
    """True when the library runs its Stream-K kernels data-parallel in THIS process (probed
    once per device); raises otherwise (a warning under ECOFLAP_ALLOW_STREAMK=1)."""
    import torch
    if not torch.cuda.is_available():
        return True
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        return True
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    if key in _verified:
        return _verified[key]
    import torch.nn.functional as F
    g = torch.Generator(device=dev).manual_seed(1234)
    rows = 2056                                   # one evaluation of the ViT-g qkv Linear: 8 x 257 tokens
    w = (torch.randn(4224, 1408, device=dev, generator=g) * 0.02).half()
    x = (torch.randn(16 * rows, 1408, device=dev, generator=g) * 0.7).half()
    with torch.no_grad():
        whole = F.linear(x, w)
        alone_first = F.linear(x[:rows].contiguous(), w)
        alone_last = F.linear(x[15 * rows:].contiguous(), w)
    ok = bool(torch.equal(whole[:rows], alone_first) and torch.equal(whole[15 * rows:], alone_last))
    del whole, alone_first, alone_last, x, w
    _verified[key] = ok
    if not ok:
        msg = (f"{ENV}=1 is not in effect in this process (environment: "
               f"{os.environ.get(ENV)!r}): hipBLASLt's Stream-K kernels are then neither "
               "reproducible call to call nor batch invariant (ecoflap_amd/blas_guard.py). "
               f"Export {ENV}=1 before the first GEMM of the process (import ecoflap_amd before "
               "running any torch matmul), or set ECOFLAP_ALLOW_STREAMK=1 to run anyway.")
        if os.environ.get("ECOFLAP_ALLOW_STREAMK") == "1":
            warnings.warn(msg)
        else:
            raise RuntimeError(msg)
    return ok
