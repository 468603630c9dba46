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


def _refuse(msg):
    if os.environ.get("ECOFLAP_ALLOW_STREAMK") == "1":
        warnings.warn(msg)
    else:
        raise RuntimeError(msg)


def _probe(dev):
    """(batch invariant at 16 evaluations, reproducible call to call) of the ViT-g qkv GEMM."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator(device=dev).manual_seed(1234)
    rows = 2056                                   # one evaluation of the ViT-g qkv Linear: 8 x 257 tokens
    w = (torch.randn(4224, 1408, device=dev, generator=g) * 0.02).half()
    x = (torch.randn(16 * rows, 1408, device=dev, generator=g) * 0.7).half()
    with torch.no_grad():
        whole = F.linear(x, w)
        again = F.linear(x, w)
        alone_first = F.linear(x[:rows].contiguous(), w)
        alone_last = F.linear(x[15 * rows:].contiguous(), w)
    invariant = bool(torch.equal(whole[:rows], alone_first)
                     and torch.equal(whole[15 * rows:], alone_last))
    repeatable = bool(torch.equal(whole, again))
    return invariant, repeatable


def verify(device=None, need_batch_invariance=True):
    """True when the library runs its Stream-K kernels data-parallel in THIS process; raises
    otherwise (a warning under ECOFLAP_ALLOW_STREAMK=1) — every time it is asked, not only the
    first: only a PASSED probe is cached per device, so a run that follows a caught refusal in
    the same process (a test session, a retry loop, a second pruner) is refused again.

    Two properties are told apart.  `TENSILE_STREAMK_DATA_PARALLEL` not being 1 in the
    environment means the kernels' hand-off of partial tiles is on: call-to-call results can
    differ (about one call in 30 000), which no single probe can show — that is refused on the
    variable alone.  Batch invariance (slots 0 and 15 of a 16-evaluation qkv GEMM equal to the
    same GEMM alone) is what the shared-suffix evaluation needs on top; a caller that never
    concatenates evaluations passes `need_batch_invariance=False` and is then not refused for a
    library version whose tile choice differs between M = 2056 and M = 32 896."""
    import torch
    if not torch.cuda.is_available():
        return True
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        return True
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    state = _verified.get(key)
    if state is None:
        state = _probe(dev)
        if all(state) and os.environ.get(ENV) == "1":
            _verified[key] = state           # successes only
    invariant, repeatable = state
    env = os.environ.get(ENV)
    if env != "1" or not repeatable:
        _refuse(f"{ENV}=1 is not in effect in this process (environment: {env!r}; probe: "
                f"batch invariant at 16 evaluations = {invariant}, same bits on a repeated call = "
                f"{repeatable}): hipBLASLt's Stream-K kernels are then not reproducible call to "
                "call (ecoflap_amd/blas_guard.py). "
                f"Export {ENV}=1 before the first GEMM of the process (import ecoflap_amd before "
                "running any torch matmul), or set ECOFLAP_ALLOW_STREAMK=1 to run anyway.")
        return False
    if not invariant and need_batch_invariance:
        _refuse(f"{ENV}=1 is in the environment, but the GEMM library is not batch invariant here "
                "(a 16-evaluation ViT-g qkv GEMM differs from the same rows alone; same bits on a "
                f"repeated call = {repeatable}).  Either the variable was set after this process's "
                "first GEMM (the library reads it once: import ecoflap_amd before any torch "
                "matmul), or this hipBLASLt picks different kernels for M = 2056 and M = 32896.  "
                "Evaluations cannot share a pass then: run with eval_batch=1, or set "
                "ECOFLAP_ALLOW_STREAMK=1 to let the loop's own per-stage probe decide.")
        return False
    if not invariant:
        warnings.warn(f"{ENV}=1 is in the environment but the batch-invariance probe failed; "
                      "this run does not concatenate evaluations, so it goes ahead — whether the "
                      "setting is live in this process could not be confirmed.")
    return True
