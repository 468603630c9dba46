#!/usr/bin/env python3
"""Is SparseGPT's factorisation chain (2 damped Cholesky + inverse; round 6: the build's own kernels,
`ECOFLAP_FACTOR_LIBRARY=1`: torch.linalg on rocSOLVER as in round 5) bit-
reproducible (a) one matrix at a time, run after run, (b) with a block's matrices side by side on
per-thread streams (`SparseGPT.factor_all`)?  Prints, per repetition that differs from the first
one-by-one result, which matrix differs and by how much."""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ecoflap_amd import hip  # noqa: E402
from ecoflap_amd.pruners.sparsegpt import SparseGPT  # noqa: E402

SHAPES = ((1408, 64, "plain"), (2048, 48, "dead"), (768, 32, "rank_deficient"), (1408, 64, "plain2"))


def build(kern):
    torch.manual_seed(11)
    out = []
    for cols, rows, kind in SHAPES:
        lin = nn.Linear(cols, rows, bias=False).cuda()
        w = SparseGPT(lin, kernels=kern)
        n_tok = 64 if kind == "rank_deficient" else 4 * cols
        x = torch.randn(n_tok, cols, device="cuda")
        if kind == "dead":
            x[:, 5:9] = 0
        w.use_mfma_hessian = False
        w.add_batch(x.unsqueeze(0), None)
        out.append(w)
    return out


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    mode = sys.argv[2] if len(sys.argv) > 2 else "both"
    kern = hip.HipKernels()
    if os.environ.get("ECOFLAP_FACTOR_LIBRARY"):
        SparseGPT.use_own_cholesky = False
        SparseGPT._own_factorisations = lambda self: True     # (force the threaded form over the library: round 5's experiment)
    ref = build(kern)
    h0 = [w.H.clone() for w in ref]
    for w in ref:
        w._factor_alone(.01)
    torch.cuda.synchronize()
    for label, side in (("one by one", False), ("side by side", True)):
        if mode not in ("both", label.split()[0]):
            continue
        bad = 0
        for r in range(reps):
            ws = build(kern)
            for w, h in zip(ws, h0):
                assert torch.equal(w.H, h), "the Hessians themselves differ"
            if side:
                SparseGPT.factor_all(ws)
            else:
                for w in ws:
                    w._factor_alone(.01)
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(zip(ref, ws)):
                if not torch.equal(a.factor[1], b.factor[1]) or not torch.equal(a.factor[0], b.factor[0]):
                    d = (a.factor[1] - b.factor[1]).abs()
                    nd = int((a.factor[1] != b.factor[1]).sum())
                    print(f"{label} rep {r}: matrix {i} {SHAPES[i]} differs in {nd} of {d.numel()} elements, max abs {float(d.max()):.3e} "
                          f"(max |Hinv| {float(a.factor[1].abs().max()):.3e}), nan {int(torch.isnan(b.factor[1]).sum())}", flush=True)
                    bad += 1
        print(f"{label}: {bad} differing results in {reps} repetitions x {len(SHAPES)} matrices", flush=True)


if __name__ == "__main__":
    main()
