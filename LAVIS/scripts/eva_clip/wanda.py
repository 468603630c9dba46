"""Uniform-sparsity Wanda on the EVA-CLIP vision tower (reference:
LAVIS/scripts/eva_clip/wanda.py:9-22; the build's ViT shape has 12 blocks)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import launch  # noqa: E402

method = "vit_wanda_pruner"
ratio = 0.5
ratios = f"{ratio}-1.0-1.0"
job_id = f"imgn-{method}_{ratios}"

sys.exit(launch("vit", (
    f"--pruning_method '{method}' --save_pruned_model"
    f" --vit_prune_spec 12-{ratios} --job_id '{job_id}'")))
