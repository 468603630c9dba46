#!/usr/bin/env python3
"""BLIP-2 shape: ECoFLaP zeroth-order table + SparseGPT local prune, end to end on the GPU
(reference: scripts/blip2/ecoflap_sparsegpt_zeroth.py; its committed run took 6801 s)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecoflap_amd import harness  # noqa: E402

args = ["--shape", "blip2", "--pruning_method", "blipt5_sparsegpt_pruner", "--score_method",
        "MEZO-GradOnly_sum", "--sparsity_ratio_granularity", "block", "--max_sparsity_per_layer", "0.7",
        "--prunining_dataset_batch_size", "1", "--num_data", "128", "--num_data_first_stage", "32",
        "--t5_prune_spec", "24-0.4-1.0-1.0", "--vit_prune_spec", "39-0.4-1.0-1.0"] + sys.argv[1:]
phases = "--phases" in args
if phases:            # device-time breakdown of stage 2 (pruners/phase_timer.py)
    args.remove("--phases")
    from ecoflap_amd.pruners.phase_timer import PhaseTimer
    PhaseTimer.enable()
t0 = time.time()
model, table = harness.main(args)
torch.cuda.synchronize()
phase_report = PhaseTimer.report() if phases else None
blocks = {k: v for k, v in model.state_dict().items() if v.dim() == 2 and ".block" in k
          and "relative_attention_bias" not in k}
zeros = sum(int((v == 0).sum()) for v in blocks.values())
total = sum(v.numel() for v in blocks.values())
wall = time.time() - t0
import hashlib  # noqa: E402
weights_sha = hashlib.sha256()          # (run-to-run reproducibility of the whole pipeline: same hash)
for k in sorted(blocks):
    weights_sha.update(blocks[k].detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
table_sha = hashlib.sha256(repr(sorted(table.items())).encode()).hexdigest() if isinstance(table, dict) else None
print(json.dumps({"wall_seconds": wall, "pruned_fraction": zeros / total,
                  "table_sha256": table_sha, "pruned_weights_sha256": weights_sha.hexdigest(),
                  "stage_stats": getattr(harness.main, "last_stage_stats", None),
                  "peak_mem_gb": torch.cuda.max_memory_allocated() / 1e9,
                  "stage2_phases": phase_report,
                  # which weight shapes ran a pinned hipBLASLt solution (shapes/fused.py)
                  "pinned_gemm": __import__("ecoflap_amd.shapes.fused", fromlist=["x"]).gemm_report()},
                 default=str))
