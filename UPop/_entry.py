"""Shared body of the four UPop entrypoints (reference: the `if not args.evaluate:` pruning
section + masked fine-tune loop of UPop/ecoflap_compress_caption.py:225-249,
ecoflap_compress_nlvr.py:233-257, ecoflap_compression_retrieval_flickr.py:351-375,
ecoflap_compression_vqa.py:250-275 and :108-129, :312-315) on shape-compatible random-init
models with synthetic task tuples.  Each entrypoint fixes (task, model shape, BERT prefix)."""
import os
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")   # before the first GEMM (ecoflap_amd/blas_guard.py)
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecoflap_amd.pruners import BLIPBertLayerWandaPruner, apply_masks_to_grads, pruning_masks  # noqa: E402
from ecoflap_amd.pruners.upop import task_forward  # noqa: E402
from ecoflap_amd.shapes import blip_bert as B  # noqa: E402
from ecoflap_amd.shapes import blip_tasks as T  # noqa: E402

TASKS = {
    # task: (bert_model_prefix, base model, toy model, batches, base image size, default batch)
    "vqa": ("text_decoder", B.blip_vqa_base, B.blip_vqa_toy,
            lambda n, bs, img, vocab, dev: B.vqa_batches(n, bs, img_size=img, vocab=vocab, device=dev),
            480, 16),
    "coco": ("text_decoder", T.blip_caption_base, T.blip_caption_toy,
             lambda n, bs, img, vocab, dev: T.caption_batches(n, bs, img_size=img, vocab=vocab, device=dev),
             384, 1),
    "nlvr": ("text_encoder", T.blip_nlvr_base, T.blip_nlvr_toy,
             lambda n, bs, img, vocab, dev: T.nlvr_batches(n, bs, img_size=img, vocab=vocab, device=dev),
             384, 1),
    "retrieval": ("text_encoder", T.blip_retrieval_base, T.blip_retrieval_toy,
                  lambda n, bs, img, vocab, dev: T.retrieval_batches(n, bs, img_size=img, vocab=vocab,
                                                                     device=dev),
                  384, 4),
}


def run(task, argv=None):
    prefix, base, toy, make_batches, img, default_bs = TASKS[task]
    ap = argparse.ArgumentParser()
    ap.add_argument("--p", type=float, default=0.5)
    ap.add_argument("--sparsity_ratio_granularity", default="block")
    ap.add_argument("--stage1", default="compat", choices=["compat", "intended"])
    ap.add_argument("--finetune_steps", type=int, default=0)
    ap.add_argument("--num_data", type=int, default=128)
    ap.add_argument("--batch_size", type=int, default=default_bs)
    ap.add_argument("--toy", action="store_true")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--save", default="")
    ap.add_argument("--z_source", default="torch", choices=["torch", "philox"],
                    help="torch (default): draw z as the reference does (torch.manual_seed + "
                         "torch.normal on the parameter's device); philox: the build's in-register "
                         "stream (opt-in)")
    args = ap.parse_args(argv)
    dev = torch.device(args.device)
    torch.manual_seed(42)
    np.random.seed(42)
    with torch.device(dev):
        model = (toy() if args.toy else base()).eval()
    loader = make_batches(args.num_data, args.batch_size, 32 if args.toy else img,
                          64 if args.toy else 30524, dev)
    t0 = time.time()
    pruner = BLIPBertLayerWandaPruner(
        model, loader,
        bert_prune_spec=f"0-{1 - args.p}-1.0-1.0", vit_prune_spec=f"0-{1 - args.p}-1.0-1.0",
        num_samples=args.num_data, bert_model_prefix=prefix, vit_model_prefix="visual_encoder",
        sparsity_ratio_granularity=args.sparsity_ratio_granularity,
        max_sparsity_per_layer=args.p + 0.1, score_method="MEZO-GradOnly_sum",
        num_data_first_stage=32, task=task, stage1_mode=args.stage1, z_source=args.z_source)
    model, table = pruner.prune()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    print(f"pruned in {time.time() - t0:.2f} s; stage stats: {pruner.stage_stats}")
    if args.save:
        torch.save(model.state_dict(), args.save)
    kept = sum(int((p != 0).sum()) for p in model.parameters())
    total = sum(p.numel() for p in model.parameters())
    print("remaining parameters", kept / total)
    if args.finetune_steps > 0:
        masks = pruning_masks(model)            # mask = (p != 0) for every named parameter
        model.train()
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5,
                                weight_decay=0.05)
        for step in range(args.finetune_steps):
            loss, _ = task_forward(task, model, loader[step % len(loader)], dev)
            opt.zero_grad()
            loss.backward()
            apply_masks_to_grads(model, masks, kernels=pruner.kernels)   # grad *= mask (K8)
            opt.step()
            print("finetune step", step, float(loss.detach()))
    return model, table
