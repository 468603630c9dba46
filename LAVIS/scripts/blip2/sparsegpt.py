"""Uniform-sparsity SparseGPT on BLIP-2 (reference: LAVIS/scripts/blip2/sparsegpt.py:9-22)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import launch  # noqa: E402

method = "blipt5_sparsegpt_pruner"
ratio = 0.5
ratios = f"{ratio}-1.0-1.0"
job_id = f"cc3m-{method}_{ratios}"

sys.exit(launch("blip2", (
    f"--pruning_method '{method}' --save_pruned_model"
    f" --t5_prune_spec 24-{ratios} --vit_prune_spec 39-{ratios} --job_id '{job_id}'")))
