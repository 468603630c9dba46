import os, sys
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from ecoflap_amd import hip
from ecoflap_amd.pruners.sparsegpt import SparseGPT
kern = hip.HipKernels()


def build():
    torch.manual_seed(11)
    out = []
    for cols, rows, kind in ((1408, 64, "plain"), (2048, 48, "dead"), (768, 32, "rank_deficient"), (1408, 64, "plain2")):
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


a, b = build(), build()
for i, (x, y) in enumerate(zip(a, b)):
    print(i, "H equal at start:", torch.equal(x.H, y.H), float(torch.diag(x.H).mean()))
Hs = [w.H.clone() for w in a]
for w in a:
    w._factor_alone(0.01)
SparseGPT.factor_all(b)
for i, (x, y) in enumerate(zip(a, b)):
    print(i, "dead equal", torch.equal(x.factor[0], y.factor[0]), "Hinv equal", torch.equal(x.factor[1], y.factor[1]),
          float(x.factor[1][0, 0]), float(y.factor[1][0, 0]), float(x.factor[1][-1, -1]), float(y.factor[1][-1, -1]))
# again, one item at a time through factor_all (len 1 -> _factor_alone) vs pairs
c = build()
SparseGPT.factor_all(c[:2]); SparseGPT.factor_all(c[2:])
for i, (x, y) in enumerate(zip(a, c)):
    print(i, "pairs: Hinv equal", torch.equal(x.factor[1], y.factor[1]), float(y.factor[1][0, 0]))
