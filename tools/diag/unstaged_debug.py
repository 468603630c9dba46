"""Which losses of the un-staged lock-step path differ from the per-evaluation (hooked) path?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
import numpy as np, torch
from ecoflap_amd import hip
from ecoflap_amd.pruners import LayerSparsity
from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss
from ecoflap_amd.pruners.losses import loss_vision_language
from ecoflap_amd.shapes import synthetic as S
from ecoflap_amd.shapes.blip2_t5 import Blip2T5, blip2_toy
from ecoflap_amd.shapes.unstaged import hide_stage_plan
hide_stage_plan(Blip2T5)
kern = hip.HipKernels()
LISTS = ["visual_encoder.blocks", "t5_model.encoder.block", "t5_model.decoder.block"]
EXTRA = ["ln_vision", "Qformer", "t5_proj"]


def run(eval_batch, verify="entries", lazy="1"):
    os.environ["ECOFLAP_LOCKSTEP_LAZY"] = lazy
    torch.manual_seed(4)
    model = blip2_toy(fp32=False).eval().to("cuda")
    batches = S.image_text_batches(16, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6, device="cuda")
    mapping = {k: ".".join(k.split(".")[:4 if k.startswith("t5") else 3])
               for k, v in model.named_parameters()
               if v.dim() == 2 and ".block" in k and "relative_attention_bias" not in k}
    loss = HookedPrefixLoss(model, loss_vision_language, LISTS, EXTRA, eval_batch=eval_batch, verify_batched=verify)
    np.random.seed(42)
    ls = LayerSparsity(model, batches, loss, 16, 0.5, 0.6, "MEZO-GradOnly_sum", 1, 1e-3, mapping,
                       kernels=kern, z_source="torch")
    ls.return_sparsity()
    torch.cuda.synchronize()
    names = [k for k, _ in model.named_parameters() if k in mapping]
    st = {k: v for k, v in loss.stats.items()}
    loss.close()
    return ls.loss_table.copy(), names, ls.seed_schedule, st


ref, names, units, _ = run(1)
for eb, verify, lazy in ((4, "entries", "1"), (4, "entries", "0"), (4, "all", "1"), (8, "entries", "1")):
    got, _, _, st = run(eb, verify, lazy)
    bad = np.argwhere(got.view(np.uint32) != ref.view(np.uint32))
    print(f"eval_batch {eb} verify {verify} lazy {lazy}: {len(bad)} of {got.size} losses differ")
    print("   stats", {k: v for k, v in st.items() if k not in ("events_total", "events_served")})
    for u, c in bad[:24]:
        li, bi, ni, seed, blen = units[u]
        print(f"   unit {u} layer {li} {names[li]} batch {bi} col {c}: {got[u, c]!r} vs {ref[u, c]!r}")
