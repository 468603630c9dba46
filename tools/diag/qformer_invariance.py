#!/usr/bin/env python3
"""Which op of the Q-Former bridge is not batch invariant?  Runs the bridge's modules on k
concatenated evaluations (batch k*8) and on each evaluation alone (batch 8) and reports, per
leaf module, whether slot i of the concatenated output equals the stand-alone output bit for bit."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

torch.manual_seed(0)
dev = "cuda"
from ecoflap_amd.shapes.blip2_t5 import Qformer  # noqa
qf = Qformer().eval().to(dev)
class _M: pass
model = _M(); model.query_tokens = torch.randn(1, 32, 768, device=dev) * 0.02
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = 8
enc = torch.randn(K * B, 257, 1408, device=dev)
q0 = model.query_tokens.expand(K * B, -1, -1).contiguous()

records = {}
def hook(name):
    def fn(mod, inp, out):
        records.setdefault(name, []).append(out.detach().clone())
    return fn
hs = []
for name, m in qf.named_modules():
    if len(list(m.children())) == 0:
        hs.append(m.register_forward_hook(hook(name)))
with torch.no_grad():
    full = qf(q0, enc)
    full_rec = {k: v[0] for k, v in records.items()}
    bad = {}
    for i in range(K):
        records.clear()
        one = qf(q0[i * B:(i + 1) * B], enc[i * B:(i + 1) * B])
        for k, v in records.items():
            a = full_rec[k]
            rows = a.shape[0] // K
            if not torch.equal(a[i * rows:(i + 1) * rows], v[0]):
                bad.setdefault(k, []).append(i)
        if not torch.equal(full[i * B:(i + 1) * B], one):
            bad.setdefault("OUTPUT", []).append(i)
shown = 0
for name, m in qf.named_modules():
    if name in bad and shown < 4:
        print(f"{name:50s} {type(m).__name__:12s} slots differing: {bad[name]}")
        shown += 1
allbad = sorted(set(i for v in bad.values() for i in v))
print(f"K={K}: modules with a difference: {len(bad)}; slots ever differing: {allbad}")
