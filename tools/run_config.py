#!/usr/bin/env python3
"""Run one BASELINE.json config end to end through the harness on the GPU and print a JSON
summary (wall time per stage, table statistics, pruned fraction).  Parity-test cases at full
size, not bench lines.

    python3 tools/run_config.py 2     # FlanT5-XL first-order GradMagAbs_sum, 128 seqs bs 1
    python3 tools/run_config.py 3     # BLIP-2 zeroth-order MEZO-GradOnly_sum, 128 pairs bs 8
"""
import os
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")   # before the first GEMM (ecoflap_amd/blas_guard.py)
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecoflap_amd import harness  # noqa: E402

CONFIGS = {
    "1": ["--shape", "vit", "--pruning_method", "vit_wanda_pruner", "--score_method", "MEZO-GradOnly_sum",
          "--sparsity_ratio_granularity", "block", "--max_sparsity_per_layer", "0.6",
          "--prunining_dataset_batch_size", "8", "--num_data", "8", "--num_data_first_stage", "8",
          "--vit_prune_spec", "12-0.5-1.0-1.0"],
    "2": ["--shape", "t5", "--pruning_method", "t5_wanda_pruner", "--score_method", "GradMagAbs_sum",
          "--sparsity_ratio_granularity", "block", "--max_sparsity_per_layer", "0.6",
          "--prunining_dataset_batch_size", "1", "--num_data", "128", "--num_data_first_stage", "128",
          "--t5_prune_spec", "24-0.5-1.0-1.0"],
    "3": ["--shape", "blip2", "--pruning_method", "blipt5_wanda_pruner", "--score_method", "MEZO-GradOnly_sum",
          "--sparsity_ratio_granularity", "block", "--max_sparsity_per_layer", "0.6",
          "--prunining_dataset_batch_size", "8", "--num_data", "128", "--num_data_first_stage", "128",
          "--t5_prune_spec", "24-0.5-1.0-1.0", "--vit_prune_spec", "39-0.5-1.0-1.0"],
    # scripts/blip2/iterative_global_gradient.py: 3 rounds, one threshold per sub-model
    "global_grad": ["--shape", "blip2", "--pruning_method", "blipt5_global_gradmagabs_pruner",
                    "--is_global", "--prune_per_model", "--iteration", "3",
                    "--prunining_dataset_batch_size", "8", "--num_data", "128",
                    "--t5_prune_spec", "24-0.5-1.0-1.0", "--vit_prune_spec", "39-0.5-1.0-1.0"],
    # scripts/blip2/mag.py
    "global_mag": ["--shape", "blip2", "--pruning_method", "blipt5_global_mag_pruner", "--is_global",
                   "--prunining_dataset_batch_size", "8", "--num_data", "128",
                   "--t5_prune_spec", "24-0.5-1.0-1.0", "--vit_prune_spec", "39-0.5-1.0-1.0"],
}


def run(which, extra=()):
    """-> summary dict of one config run through the harness (tests/test_full_configs.py)"""
    extra = list(extra)
    t0 = time.time()
    model, table = harness.main(CONFIGS[which] + extra)
    torch.cuda.synchronize()
    wall = time.time() - t0
    blocks = {k: v for k, v in model.state_dict().items() if v.dim() == 2 and ".block" in k
              and "relative_attention_bias" not in k}
    zeros = sum(int((v == 0).sum()) for v in blocks.values())
    total = sum(v.numel() for v in blocks.values())
    vals = sorted(set(round(v, 6) for v in table.values())) if isinstance(table, dict) else []
    import hashlib
    dump = os.environ.get("ECOFLAP_DUMP_LOSS_TABLE")      # stage 1's [units, 2] fp32 losses, for run-to-run diffs
    if dump and getattr(harness.main, "last_loss_table", None) is not None:
        import numpy as np
        np.save(dump, np.asarray(harness.main.last_loss_table))
    table_sha = (hashlib.sha256(repr(sorted(table.items())).encode()).hexdigest()
                 if isinstance(table, dict) else None)
    weights_sha = hashlib.sha256()
    for k in sorted(blocks):
        weights_sha.update(blocks[k].detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    return {
        "config": which, "wall_seconds": wall, "prunable_matrices": len(blocks),
        "prunable_elements": total, "pruned_fraction": zeros / total,
        "table_entries": len(table) if isinstance(table, dict) else 0,
        "table_sha256": table_sha, "pruned_weights_sha256": weights_sha.hexdigest(),
        "distinct_sparsities": len(vals), "min_sparsity": vals[0] if vals else None,
        "max_sparsity": vals[-1] if vals else None,
        "peak_mem_gb": torch.cuda.max_memory_allocated() / 1e9,
        "stage_stats": getattr(harness.main, "last_stage_stats", None),
        "pinned_gemm": __import__("ecoflap_amd.shapes.fused", fromlist=["x"]).gemm_report()}


def main():
    print(json.dumps(run(sys.argv[1], sys.argv[2:]), default=str))


if __name__ == "__main__":
    main()
