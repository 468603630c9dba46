"""Perturbation-size sweep of the zeroth-order stage on BLIP-2 (reference:
LAVIS/scripts/blip2/ecoflap_zeroth_eps.py:9-32; its `olmezo-gradient_sum` is the pre-release name
of `MEZO-GradOnly_sum`, which is what the shipped LayerSparsity understands)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import launch  # noqa: E402

method = "blipt5_wanda_pruner"
sparsity_ratio_granularity = "block"
score_method = "MEZO-GradOnly_sum"
ratio = 0.5
ratios = f"{ratio}-1.0-1.0"
max_sparsity_per_layer = f"{round(1.0 - ratio + 0.1, 1)}"

rc = 0
for noise_eps in [1e-1, 1e-2, 1e-4]:
    job_id = (f"cc3m-{method}_{ratios}_{score_method}{max_sparsity_per_layer}"
              f"_{sparsity_ratio_granularity}_eps{noise_eps}")
    rc |= launch("blip2", (
        f"--pruning_method '{method}' --save_pruned_model"
        f" --score_method {score_method}"
        f" --max_sparsity_per_layer {max_sparsity_per_layer}"
        f" --sparsity_ratio_granularity {sparsity_ratio_granularity}"
        f" --noise_eps {noise_eps}"
        f" --t5_prune_spec 24-{ratios} --vit_prune_spec 39-{ratios} --job_id '{job_id}'"))
sys.exit(rc)
