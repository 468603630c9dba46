"""ECoFLaP (first-order table) + SparseGPT on BLIP-2 (reference:
LAVIS/scripts/blip2/ecoflap_sparsegpt_first.py:10-24 — there the table comes from a previous
ecoflap_first run through --sparsity_dict; here stage 1 runs in the same job unless a table is
passed as the third argument)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import launch  # noqa: E402

method = "blipt5_sparsegpt_pruner"
ratio = 0.4
ratios = f"{ratio}-1.0-1.0"
job_id = f"cc3m-{method}_{ratios}_aobd_sum0.7_block"
table = f" --sparsity_dict {sys.argv.pop(3)}" if len(sys.argv) > 3 else (
    " --score_method GradMagAbs_sum --sparsity_ratio_granularity block --max_sparsity_per_layer 0.7"
    " --num_data_first_stage 128")

sys.exit(launch("blip2", (
    f"--pruning_method '{method}' --save_pruned_model{table}"
    f" --t5_prune_spec 24-{ratios} --vit_prune_spec 39-{ratios} --job_id '{job_id}'")))
