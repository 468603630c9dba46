"""Pruning harness — the build's counterpart of the reference's `evaluate_{blip,t5,eva_clip}.py`
for this path only (SURVEY.md §8 a-H): CLI flags with the reference's names and defaults
(LAVIS/evaluate_blip.py:37-284), the config dict handed to `load_pruner`
(LAVIS/evaluate_blip.py:399-418), and the three output files
(`pruned_checkpoint/{job_id}.pth`, `sparsity_dict/{job_id}.yaml`,
`training_statistics/{job_id}.yaml`, LAVIS/evaluate_blip.py:438-472).

Datasets, tasks, runners and checkpoints are out of scope: the model is a shape-compatible
random-init module and the calibration loader is synthetic (`--shape`, `--toy`).

    python -m ecoflap_amd.harness --shape blip2 --pruning_method blipt5_wanda_pruner \
        --score_method MEZO-GradOnly_sum --sparsity_ratio_granularity block \
        --max_sparsity_per_layer 0.6 --prunining_dataset_batch_size 8 \
        --t5_prune_spec 24-0.5-1.0-1.0 --vit_prune_spec 39-0.5-1.0-1.0 --save_pruned_model \
        --job_id demo
"""
import os
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")   # before the first GEMM (ecoflap_amd/blas_guard.py)
import argparse
import os
import time

import numpy as np
import torch
import yaml


def build_parser():
    p = argparse.ArgumentParser(description="ECoFLaP pruning on MI355X (hot path only)")
    p.add_argument("--shape", default="blip2", choices=["blip2", "t5", "vit"],
                   help="blip2: EVA-ViT-g + Q-Former + FlanT5-XL; t5: FlanT5-XL; vit: ViT-B/16 CLIP shape")
    p.add_argument("--toy", action="store_true", help="CPU-sized shapes of the same families")
    p.add_argument("--device", default="cuda")
    # the reference's flags (names, types, defaults)
    p.add_argument("--pruning_method", type=str, default=None)
    p.add_argument("--save_pruned_model", action="store_true")
    p.add_argument("--job_id", type=str, default="job")
    p.add_argument("--t5_prune_spec", type=str, default=None)
    p.add_argument("--vit_prune_spec", type=str, default=None)
    p.add_argument("--num_data", type=int, default=128)
    p.add_argument("--prunining_dataset_batch_size", type=int, default=1)
    p.add_argument("--is_global", action="store_true")
    p.add_argument("--sparsity_ratio_granularity", type=str, default=None)
    p.add_argument("--max_sparsity_per_layer", type=float, default=0.8)
    p.add_argument("--score_method", type=str, default="obd_avg")
    p.add_argument("--num_data_first_stage", type=int, default=32)
    p.add_argument("--num_noise", default=1, type=int)
    p.add_argument("--noise_eps", default=1e-3, type=float)
    p.add_argument("--sparsity_dict", type=str, default=None)
    p.add_argument("--prune_per_model", action="store_true")
    p.add_argument("--iteration", type=int, default=1)
    p.add_argument("--t5_pruned_checkpoint", type=str, default=None)
    p.add_argument("--vit_pruned_checkpoint", type=str, default=None)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--out_dir", type=str, default=".")
    # build-side: where the zeroth-order perturbation z comes from
    p.add_argument("--z_source", default="torch", choices=["torch", "philox"],
                   help="torch (default): torch.manual_seed(seed) + torch.normal on the parameter's "
                        "device, the reference's own draw (layer_single_base_pruner.py:482-485): the "
                        "table the reference's arithmetic gives on this GPU; "
                        "philox: z generated in registers by the K1 kernel, never in memory — the "
                        "build's own stream (opt-in: no reference run can equal its table)")
    p.add_argument("--k1_form", default="block", choices=["block", "units", "triple", "single"],
                   help="K1 launch form (bit-identical results): one launch per transformer block "
                        "(default), per layer, per unit, or the reference's three in-place passes")
    p.add_argument("--eval_batch", type=int, default=0,
                   help="loss evaluations of a layer per pass of the batch-invariant suffix (1: one "
                        "suffix per evaluation; 0, the default: sized from the calibration set — 16 "
                        "to 64, all of a layer's evaluations in one pass up to ~256 samples)")
    p.add_argument("--lanes", type=int, default=2, help="concurrent evaluation lanes (weight replicas)")
    p.add_argument("--unstaged", action="store_true",
                   help="hide the shape module's stage_plan(): the pruners then see what a reference "
                        "user's own model looks like (block lists and a forward, INTEGRATION.md §A) and "
                        "score it through pruners/hooked_prefix.py — same table, same pruned weights")
    p.add_argument("--stage1_checkpoint", type=str, default=None,
                   help="zeroth-order stage 1: file that receives the loss table of the finished "
                        "layers every 32 layers; a rerun of the same command resumes behind the last "
                        "saved layer and ends with the same table and weights (the reference has no "
                        "mid-stage-1 resume)")
    return p


def build_model_and_loader(args, device):
    from .shapes import synthetic as S
    bs = args.prunining_dataset_batch_size
    if getattr(args, "unstaged", False):
        from .shapes.blip2_t5 import Blip2T5
        from .shapes.eva_clip import EVACLIP
        from .shapes.t5 import T5 as _T5
        from .shapes.unstaged import hide_stage_plan
        hide_stage_plan(Blip2T5, _T5, EVACLIP)
    if args.shape == "blip2":
        from .shapes.blip2_t5 import blip2_flant5xl, blip2_toy
        with torch.device(device):
            model = blip2_toy(fp32=(device.type == "cpu")) if args.toy else blip2_flant5xl()
        loader = S.image_text_batches(args.num_data, bs, img_size=28 if args.toy else 224,
                                      vocab=96 if args.toy else 32128, seed=args.seed,
                                      device=device)
    elif args.shape == "t5":
        from .shapes.t5 import T5, t5_config
        cfg = (t5_config(d_model=32, d_kv=8, num_heads=4, d_ff=64, num_layers=2, vocab_size=96)
               if args.toy else t5_config())
        with torch.device(device):
            model = T5(cfg, dtype=None if device.type == "cpu" else torch.bfloat16,
                       init_std=0.2 if args.toy else 0.02)
        loader = S.text_batches(args.num_data, bs, vocab=96 if args.toy else 32128, seed=args.seed,
                                device=device)
    else:
        from .shapes.eva_clip import vit_b16_clip, vit_toy
        with torch.device(device):
            model = vit_toy() if args.toy else vit_b16_clip()
        loader = S.image_label_batches(args.num_data, bs, img_size=32 if args.toy else 224,
                                       num_classes=5 if args.toy else 1000, seed=args.seed,
                                       device=device)
    return model.eval(), loader


def load_pruned_checkpoints(model, t5_pruned_checkpoint=None, vit_pruned_checkpoint=None):
    """Re-ingest `pruned_checkpoint/<job>.pth` files the way the evaluation runs of every
    launcher do (LAVIS/evaluate_blip.py:345-390): the T5 part by its `t5_model.` prefix, strict;
    the ViT part from either a BLIP-2 (`visual_encoder.`) or an EVA-CLIP (`visual.`) checkpoint,
    overriding only the keys the target owns.  (`interpolate_pos_embed` is the model zoo's and
    is a no-op at equal image size; differing sizes are rejected here.)"""
    if t5_pruned_checkpoint is not None and getattr(model, "t5_model", None) is not None:
        print("Load t5 pruned weight")
        sd = torch.load(t5_pruned_checkpoint, map_location="cpu")
        sd = {k.replace("t5_model.", ""): v for k, v in sd.items() if k.startswith("t5_model")}
        model.t5_model.load_state_dict(sd)
    if vit_pruned_checkpoint is not None:
        print("Load vit pruned weight")
        sd = torch.load(vit_pruned_checkpoint, map_location="cpu")
        model_prefix = None
        for candidate in ["visual.", "visual_encoder."]:
            if any(k.startswith(candidate) for k in sd.keys()):
                model_prefix = candidate
                break
        assert model_prefix is not None
        print(f"VIT checkpoint prefix: {model_prefix}")
        sd = {k.replace(model_prefix, ""): v for k, v in sd.items() if k.startswith(model_prefix)}
        target = model.visual_encoder.state_dict()
        for k, v in sd.items():
            if k in target:
                if tuple(v.shape) != tuple(target[k].shape):
                    raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != model "
                                     f"{tuple(target[k].shape)} (position-embedding interpolation "
                                     "belongs to the model zoo, not to this path)")
                target[k] = v
        model.visual_encoder.load_state_dict(target)
    return model


def checkpoint_to_save(state_dict, shape):
    """What `--save_pruned_model` writes: the whole `state_dict` for BLIP-2 and T5
    (LAVIS/evaluate_blip.py:442-445, evaluate_t5.py:371); for the EVA-CLIP entry point only the
    vision tower's keys, and of those none of `blocks.39` — the block EVA-ViT-g drops when it
    serves as BLIP-2's encoder (LAVIS/evaluate_eva_clip.py:414-424: substring tests, as there)."""
    if shape != "vit":
        return state_dict
    return {k: v for k, v in state_dict.items() if "blocks.39" not in k and "visual." in k}


def config_dict(args):
    """Keys and values of LAVIS/evaluate_blip.py:399-418 (t5/eva_clip variants use prune_spec)."""
    cfg = {
        "importance_scores_cache": None,
        "keep_indices_cache": None,
        "is_strct_pruning": False,
        "is_global": args.is_global,
        "num_samples": args.num_data,
        "sparsity_ratio_granularity": args.sparsity_ratio_granularity,
        "max_sparsity_per_layer": args.max_sparsity_per_layer,
        "score_method": args.score_method,
        "num_data_first_stage": args.num_data_first_stage,
        "num_noise": args.num_noise,
        "noise_eps": args.noise_eps,
        "sparsity_dict": args.sparsity_dict,
        "prune_per_model": args.prune_per_model,
        "iteration": args.iteration,
        "z_source": getattr(args, "z_source", "torch"),      # build-side extras (kw-only)
        "k1_form": getattr(args, "k1_form", "block"),
        "eval_batch": getattr(args, "eval_batch", 0),
        "n_lanes": getattr(args, "lanes", 2),
        "stage1_checkpoint": getattr(args, "stage1_checkpoint", None),
    }
    if str(args.pruning_method).startswith("blipt5_"):
        cfg.update(t5_prune_spec=args.t5_prune_spec, vit_prune_spec=args.vit_prune_spec,
                   t5_pruning_method="none", vit_pruning_method="none")
    elif str(args.pruning_method).startswith("t5_"):
        cfg.update(prune_spec=args.t5_prune_spec)
    else:
        cfg.update(prune_spec=args.vit_prune_spec)
    return cfg


def main(argv=None, kernels=None):
    """kernels: backend object for the pruner (tests hand in the oracle-backed checker; None =
    the HIP library, which raises if it is not built)."""
    args = build_parser().parse_args(argv)
    from . import load_pruner
    device = torch.device(args.device)
    # setup_seeds(seed + rank), LAVIS/evaluate_blip.py:287-295
    import random
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    model, loader = build_model_and_loader(args, device)
    load_pruned_checkpoints(model, args.t5_pruned_checkpoint, args.vit_pruned_checkpoint)
    # the EVA-CLIP entry point counts and saves the vision tower only
    # (LAVIS/evaluate_eva_clip.py:334-336, :404-424)
    counted = model.visual if args.shape == "vit" else model
    orig_total = sum((p != 0).float().sum() for p in counted.parameters())
    cfg = config_dict(args)
    if kernels is not None:
        cfg["kernels"] = kernels
    pruner = load_pruner(args.pruning_method, model, loader, cfg=cfg)
    start = time.time()
    model, sparsity_dict = pruner.prune()
    remaining = sum((p != 0).float().sum() for p in counted.parameters())
    print(float(remaining / orig_total * 100))
    if args.save_pruned_model:
        for folder in ("pruned_checkpoint", "sparsity_dict", "training_statistics"):
            os.makedirs(os.path.join(args.out_dir, folder), exist_ok=True)
        torch.save(checkpoint_to_save(model.state_dict(), args.shape),
                   os.path.join(args.out_dir, "pruned_checkpoint", args.job_id + ".pth"))
        if sparsity_dict is not None and isinstance(sparsity_dict, dict):
            with open(os.path.join(args.out_dir, "sparsity_dict", args.job_id + ".yaml"), "w") as f:
                yaml.dump(sparsity_dict, f)
        peak = (torch.cuda.max_memory_allocated() / 1024 ** 2) / 1000 if device.type == "cuda" else 0.0
        with open(os.path.join(args.out_dir, "training_statistics", args.job_id + ".yaml"), "w") as f:
            yaml.dump({"memory": peak, "time": time.time() - start}, f)
    main.last_stage_stats = dict(getattr(pruner, "stage_stats", {}))
    main.last_loss_table = getattr(pruner, "last_loss_table", None)
    return model, sparsity_dict


if __name__ == "__main__":
    main()
