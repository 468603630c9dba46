"""Per-event comparison of the lock-step path against the sequential path for the offending chunk."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
os.environ["ECOFLAP_LOCKSTEP_LAZY"] = "0"
import numpy as np, torch
from ecoflap_amd import hip
from ecoflap_amd.pruners import LayerSparsity
from ecoflap_amd.pruners import hooked_prefix as HP
from ecoflap_amd.pruners.losses import loss_vision_language
from ecoflap_amd.shapes import synthetic as S
from ecoflap_amd.shapes.blip2_t5 import Blip2T5, blip2_toy
from ecoflap_amd.shapes.unstaged import hide_stage_plan
hide_stage_plan(Blip2T5)
kern = hip.HipKernels()
LISTS = ["visual_encoder.blocks", "t5_model.encoder.block", "t5_model.decoder.block"]
EXTRA = ["ln_vision", "Qformer", "t5_proj"]
TRACE = {"on": False, "lock": {}, "seq": {}, "args_lock": {}, "args_seq": {}}


def first_tensor(o):
    return [t for t in HP._flatten(o)[0] if torch.is_tensor(t)]


class Dbg(HP.HookedPrefixLoss):
    def _on_event(self, ctx, mod, real, args, kwargs):
        i = ctx.counter
        out = super()._on_event(ctx, mod, real, args, kwargs)
        if TRACE["on"]:
            which = "lock" if ctx.lock else "seq"
            slot = ctx.slot if ctx.lock else TRACE.get("slot", 0)
            TRACE[which][(slot, i)] = [t.detach().clone() for t in first_tensor(out)]
            TRACE["args_" + which][(slot, i)] = [t.detach().clone() for t in first_tensor((args, kwargs))]
        return out

    def _lockstep(self, model, evals, cuda_enabled):
        hot = self._pair_name.endswith("blocks.0.mlp.fc2.weight")
        if hot:
            TRACE["on"] = True
            TRACE["lock"].clear(); TRACE["seq"].clear()
        losses = super()._lockstep(model, evals, cuda_enabled)
        if not hot or losses is None:
            TRACE["on"] = False
            return losses
        want = []
        for s, ev in enumerate(evals):
            TRACE["slot"] = s
            want += self._sequential(model, [ev], cuda_enabled)
        TRACE["on"] = False
        bad = [i for i in range(len(evals)) if not torch.equal(losses[i], want[i])]
        print("chunk", [float(x) for x in losses], [float(x) for x in want], "bad", bad, flush=True)
        if bad:
            names = [self.paths[m] for m in self.sequence]
            for (slot, i), outs in sorted(TRACE["lock"].items()):
                ref = TRACE["seq"].get((slot, i))
                if ref is None:
                    print("slot", slot, "event", i, names[i], "no sequential record"); continue
                same = len(ref) == len(outs) and all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(outs, ref))
                a_l, a_s = TRACE["args_lock"].get((slot, i)), TRACE["args_seq"].get((slot, i))
                same_args = a_l is not None and a_s is not None and len(a_l) == len(a_s) and all(
                    x.shape == y.shape and torch.equal(x, y) for x, y in zip(a_l, a_s))
                if not same or not same_args:
                    d = [float((a.float() - b.float()).abs().max()) for a, b in zip(outs, ref) if a.shape == b.shape]
                    print("   slot", slot, "event", i, names[i], "outputs equal", same, "args equal", same_args, "max diff", d)
        return want


torch.manual_seed(4)
model = blip2_toy(fp32=False).eval().to("cuda")
batches = S.image_text_batches(16, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6, device="cuda")
mapping = {k: ".".join(k.split(".")[:4 if k.startswith("t5") else 3])
           for k, v in model.named_parameters()
           if v.dim() == 2 and ".block" in k and "relative_attention_bias" not in k}
keep = {k: v for k, v in mapping.items() if k.startswith("visual_encoder.blocks.0")}
loss = Dbg(model, loss_vision_language, LISTS, EXTRA, eval_batch=4)
np.random.seed(42)
ls = LayerSparsity(model, batches, loss, 16, 0.5, 0.6, "MEZO-GradOnly_sum", 1, 1e-3, keep,
                   kernels=kern, z_source="torch")
ls.compute_importance_scores_mezo(keep)
names = [loss.paths[m] for m in loss.sequence]
for (slot, i), outs in sorted(TRACE["lock"].items()):
    ref = TRACE["seq"].get((slot, i))
    if ref is None:
        print("slot", slot, "event", i, names[i], "no sequential record"); continue
    same = len(ref) == len(outs) and all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(outs, ref))
    a_l, a_s = TRACE["args_lock"].get((slot, i)), TRACE["args_seq"].get((slot, i))
    same_args = a_l is not None and a_s is not None and len(a_l) == len(a_s) and all(
        x.shape == y.shape and torch.equal(x, y) for x, y in zip(a_l, a_s))
    if not same or not same_args:
        print("slot", slot, "event", i, names[i], "outputs equal", same, "args equal", same_args)
print("done", len(TRACE["lock"]), len(TRACE["seq"]))
