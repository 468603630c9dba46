"""factor_all against _factor_alone inside a real SparseGPT run (uniform ratios: no stage 1)."""
import os, sys
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ecoflap_amd import harness
from ecoflap_amd.pruners.sparsegpt import SparseGPT

real = SparseGPT.factor_all.__func__
count = [0]


def checked(cls, items, percdamp=.01):
    items = [it for it in items if it.factor is None]
    for it in items:
        it.flush()
    copies = [it.H.clone() for it in items]
    try:
        real(cls, items, percdamp)
    except Exception as ex:
        print("factor_all raised:", ex, flush=True)
        for i, H in enumerate(copies):
            print("  item", i, tuple(H.shape), "nan", bool(torch.isnan(H).any()), "inf", bool(torch.isinf(H).any()),
                  "diag mean", float(torch.diag(H).mean()), "diag min", float(torch.diag(H).min()), flush=True)
    for i, (it, H) in enumerate(zip(items, copies)):
        alone = SparseGPT.__new__(SparseGPT)
        alone.H, alone.factor = H, None
        try:
            alone._factor_alone(percdamp)
        except Exception as ex:
            print("  alone raised for item", i, ex, flush=True)
            continue
        if it.factor is None:
            print("  item", i, "no factor from factor_all; alone ok", flush=True)
            it.factor = alone.factor
            continue
        same = torch.equal(it.factor[1], alone.factor[1])
        if not same:
            print(f"  block call {count[0]} item {i} {tuple(H.shape)}: Hinv differs, max diff",
                  float((it.factor[1] - alone.factor[1]).abs().max()), "alone[0,0]", float(alone.factor[1][0, 0]),
                  "batched[0,0]", float(it.factor[1][0, 0]), flush=True)
            it.factor = alone.factor
    count[0] += 1


SparseGPT.factor_all = classmethod(checked)
args = ["--shape", "blip2", "--pruning_method", "blipt5_sparsegpt_pruner", "--prunining_dataset_batch_size", "1",
        "--num_data", "32", "--t5_prune_spec", "24-0.5-1.0-1.0", "--vit_prune_spec", "39-0.5-1.0-1.0"]
model, table = harness.main(args)
torch.cuda.synchronize()
print("done; factor_all calls", count[0])
