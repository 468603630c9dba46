"""ECoFLaP zeroth-order + Wanda on the CLIP vision tower (reference: LAVIS/scripts/eva_clip/ecoflap.py:10-31;
the build's synthetic shape is ViT-B/16 with EVA parameter names, 12 blocks)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import launch  # noqa: E402

method = "vit_wanda_pruner"
sparsity_ratio_granularity = "block"
score_method = "MEZO-GradOnly_sum"
ratio = 0.5
ratios = f"{ratio}-1.0-1.0"
max_sparsity_per_layer = f"{round(1.0 - ratio + 0.1, 1)}"
prunining_dataset_batch_size = 8
job_id = (f"imgn-{method}_{ratios}_{score_method}{max_sparsity_per_layer}"
          f"_{sparsity_ratio_granularity}_bs{prunining_dataset_batch_size}")

sys.exit(launch("vit", (
    f"--pruning_method '{method}' --save_pruned_model"
    f" --score_method {score_method}"
    f" --sparsity_ratio_granularity {sparsity_ratio_granularity}"
    f" --max_sparsity_per_layer {max_sparsity_per_layer}"
    f" --prunining_dataset_batch_size {prunining_dataset_batch_size}"
    f" --vit_prune_spec 12-{ratios} --job_id '{job_id}'")))
