"""Narrow down the one wrong lock-step loss: re-run the offending chunk, with variations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
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


class Dbg(HP.HookedPrefixLoss):
    def _lockstep(self, model, evals, cuda_enabled):
        self._verified.add(("skip",))
        save = self.verify_batched
        losses = super()._lockstep(model, evals, cuda_enabled)
        if losses is None:
            return None
        want = self._sequential(model, evals, cuda_enabled)
        bad = [i for i in range(len(evals)) if not torch.equal(losses[i], want[i])]
        if bad:
            print("MISMATCH", self._pair_name, "slots", bad, [float(x) for x in losses], [float(x) for x in want],
                  "owner_ok", self._owner_ok.get((list(self._owner_ok)[0][0], self._pair_name)) if self._owner_ok else None,
                  flush=True)
            for rep in range(2):
                again = super()._lockstep(model, evals, cuda_enabled)
                print("   again", [float(x) for x in again], flush=True)
            os.environ["DBG_SYNC"] = "1"
            again = super()._lockstep(model, evals, cuda_enabled)
            print("   with syncs", [float(x) for x in again], flush=True)
            os.environ.pop("DBG_SYNC")
        return want

    def _owner_event(self, e, mod, spec, cat, per, thetas, fam, B, k):
        if os.environ.get("DBG_SYNC"):
            torch.cuda.synchronize()
        out = super()._owner_event(e, mod, spec, cat, per, thetas, fam, B, k)
        if os.environ.get("DBG_SYNC"):
            torch.cuda.synchronize()
        return out


torch.manual_seed(4)
model = blip2_toy(fp32=False).eval().to("cuda")
batches = S.image_text_batches(16, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6, device="cuda")
mapping = {k: ".".join(k.split(".")[:4 if k.startswith("t5") else 3])
           for k, v in model.named_parameters()
           if v.dim() == 2 and ".block" in k and "relative_attention_bias" not in k}
loss = Dbg(model, loss_vision_language, LISTS, EXTRA, eval_batch=4)
np.random.seed(42)
ls = LayerSparsity(model, batches, loss, 16, 0.5, 0.6, "MEZO-GradOnly_sum", 1, 1e-3, mapping,
                   kernels=kern, z_source="torch")
ls.return_sparsity()
print({k: v for k, v in loss.stats.items() if k not in ("events_total", "events_served")})
