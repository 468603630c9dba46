"""Does an op's result depend on the THREAD that issues it (per-thread library handles)?"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
import torch
from ecoflap_amd.shapes import synthetic as S
from ecoflap_amd.shapes.blip2_t5 import blip2_toy, blip2_flant5xl

for name, build, img in (("toy", lambda: blip2_toy(fp32=False), 28), ("full", blip2_flant5xl, 224)):
    torch.manual_seed(4)
    with torch.device("cuda"):
        model = build().eval()
    batches = S.image_text_batches(16, 2 if name == "toy" else 8, img_size=img, vocab=96 if name == "toy" else 32128,
                                   in_len=5, out_len=4, seed=6, device="cuda")
    vit = model.visual_encoder

    def embed(b):
        with torch.no_grad(), model.maybe_autocast():
            return vit.embed(b["image"])

    def full(b):
        with torch.no_grad():
            return model(b)["loss"]

    for what, fn in (("vit.embed", embed), ("whole forward loss", full)):
        main = [fn(b).clone() for b in batches]
        res = {}

        def work(t):
            res[t] = [fn(b).clone() for b in batches]
        ths = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for t in ths:
            t.start(); t.join()
        again = [fn(b).clone() for b in batches]
        torch.cuda.synchronize()
        for t in range(4):
            bad = [i for i in range(len(batches)) if not torch.equal(res[t][i], main[i])]
            print(name, what, "thread", t, "differs from main on batches", bad,
                  [float((res[t][i].float() - main[i].float()).abs().max()) for i in bad][:4])
        print(name, what, "main again differs", [i for i in range(len(batches)) if not torch.equal(again[i], main[i])])
    del model
    torch.cuda.empty_cache()
