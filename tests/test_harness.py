"""Harness (a-H): reference flag names -> cfg dict -> load_pruner -> output files."""
import os

import torch
import yaml


def test_config_dict_keys_match_reference():
    from ecoflap_amd.harness import build_parser, config_dict
    args = build_parser().parse_args([
        "--pruning_method", "blipt5_wanda_pruner", "--score_method", "MEZO-GradOnly_sum",
        "--sparsity_ratio_granularity", "block", "--max_sparsity_per_layer", "0.6",
        "--t5_prune_spec", "24-0.5-1.0-1.0", "--vit_prune_spec", "39-0.5-1.0-1.0"])
    cfg = config_dict(args)
    # LAVIS/evaluate_blip.py:399-418
    assert list(cfg.keys())[:14] == [
        "importance_scores_cache", "keep_indices_cache", "is_strct_pruning", "is_global",
        "num_samples", "sparsity_ratio_granularity", "max_sparsity_per_layer", "score_method",
        "num_data_first_stage", "num_noise", "noise_eps", "sparsity_dict", "prune_per_model",
        "iteration"]
    assert cfg["num_data_first_stage"] == 32 and cfg["noise_eps"] == 1e-3 and cfg["num_noise"] == 1
    assert cfg["t5_pruning_method"] == "none" and cfg["vit_pruning_method"] == "none"


def test_launcher_names_exist():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rel in ["LAVIS/scripts/blip2/ecoflap_zeroth.py", "LAVIS/scripts/blip2/ecoflap_first.py",
                "LAVIS/scripts/t5/ecoflap.py", "LAVIS/scripts/eva_clip/ecoflap.py"]:
        assert os.path.exists(os.path.join(root, rel)), rel


def test_harness_writes_reference_outputs(tmp_path, monkeypatch):
    """Runs end to end on CPU with the checker backend injected (the product backend is HIP-only)."""
    from oracle_backend import OracleKernels, torch_cpu_normal
    import ecoflap_amd.harness as H
    import ecoflap_amd
    real = ecoflap_amd.load_pruner

    def with_checker(name, model, loader, cfg_path=None, cfg=None):
        return real(name, model, loader, cfg=dict(cfg, kernels=OracleKernels(), z_source=torch_cpu_normal))

    monkeypatch.setattr(ecoflap_amd, "load_pruner", with_checker)
    model, table = H.main([
        "--shape", "vit", "--toy", "--device", "cpu", "--pruning_method", "vit_wanda_pruner",
        "--score_method", "MEZO-GradOnly_sum", "--sparsity_ratio_granularity", "block",
        "--max_sparsity_per_layer", "0.6", "--prunining_dataset_batch_size", "2", "--num_data", "8",
        "--num_data_first_stage", "8", "--vit_prune_spec", "3-0.5-1.0-1.0", "--save_pruned_model",
        "--job_id", "t", "--out_dir", str(tmp_path)])
    sd = torch.load(tmp_path / "pruned_checkpoint" / "t.pth")
    assert set(sd.keys()) == set(model.state_dict().keys())
    saved = yaml.safe_load(open(tmp_path / "sparsity_dict" / "t.yaml"))
    assert saved == table and len(saved) == 12
    stats = yaml.safe_load(open(tmp_path / "training_statistics" / "t.yaml"))
    assert set(stats) == {"memory", "time"}
    zeros = sum(int((v == 0).sum()) for k, v in sd.items() if ".blocks." in k and v.dim() == 2)
    total = sum(v.numel() for k, v in sd.items() if ".blocks." in k and v.dim() == 2)
    assert 0.45 < zeros / total < 0.56
