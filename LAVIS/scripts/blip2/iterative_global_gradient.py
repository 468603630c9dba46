"""Iterative global |grad|*|W| baseline on BLIP-2, three rounds, one threshold per sub-model
(reference: LAVIS/scripts/blip2/iterative_global_gradient.py:9-28)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import launch  # noqa: E402

method = "blipt5_global_gradmagabs_pruner"
ratio = 0.5
ratios = f"{ratio}-1.0-1.0"
iteration = 3
job_id = f"cc3m-{method}_{ratios}_iteration{iteration}_global_per_model"

sys.exit(launch("blip2", (
    f"--pruning_method '{method}' --save_pruned_model --is_global --prune_per_model"
    f" --iteration {iteration}"
    f" --t5_prune_spec 24-{ratios} --vit_prune_spec 39-{ratios} --job_id '{job_id}'")))
