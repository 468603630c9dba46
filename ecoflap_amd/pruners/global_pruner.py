"""Global (unstructured) baselines of the BLIP-2 launchers `scripts/blip2/mag.py` and
`scripts/blip2/iterative_global_gradient.py`: `blipt5_global_mag_pruner`,
`blipt5_global_gradmagabs_pruner`, `blipt5_global_mezo_pruner`
(LAVIS/lavis/compression/pruners/global_pruner.py:56-389).

Same iteration as `LayerSparsity.global_iterative_pruning`, but the pruned weights are the
result (nothing is restored, `prune()` returns `(model, None)`), and the threshold is taken
over all layers (`--is_global`), per sub-model (`--is_global --prune_per_model`) or per layer
(neither).  Scores are never materialised: `ecoflap_global_threshold_prune` recomputes them
from (W, |g| accumulator, mask) inside its histogram passes; one call = one threshold.

As shipped, kept: the "magnitude" pruner scores the SIGNED weight (`v.data.float()`, :251), so
it removes the most negative weights first; the MeZO variant has one score per layer, so its
masks broadcast and whole matrices are zeroed (:385-387, :200)."""
import time

import torch

from ..registry import registry
from .base_pruner import LayerWiseBasePruner, print_time
from .layer_sparsity import LayerSparsity
from .losses import loss_vision_language


class BLIPT5GlobalPruner(LayerWiseBasePruner):
    pruner_name = "blipt5_global_pruner"
    score_mode = None          # ecoflap_global_threshold_prune mode; None = per-layer scalars

    def __init__(self, model, data_loader, t5_prune_spec=None, vit_prune_spec=None,
                 t5_pruning_method=None, vit_pruning_method=None, t5_model_prefix="t5_model",
                 vit_model_prefix="visual_encoder", iteration=1, **kwargs):
        kwargs.pop("prune_spec", None)
        kwargs.pop("model_prefix", None)
        super().__init__(model=model, data_loader=data_loader, prune_spec=None,
                         model_prefix="tmp", **kwargs)
        self.t5_prune_spec = t5_prune_spec
        self.vit_prune_spec = vit_prune_spec
        self.t5_model_prefix = t5_model_prefix
        self.vit_model_prefix = vit_model_prefix
        self.iteration = iteration

    # ------------------------------------------------------------------ helpers
    def _kernels(self):
        if self.kernels is None:
            from .. import hip
            self.kernels = hip.HipKernels()
        return self.kernels

    def _threshold_groups(self, names):
        """Index lists that share one threshold (:179-195)."""
        if self.is_global and not self.prune_per_model:
            print("global")
            return [list(range(len(names)))]
        if self.is_global and self.prune_per_model:
            print("model-level global")
            return [[i for i, k in enumerate(names) if k.startswith(self.vit_model_prefix)],
                    [i for i, k in enumerate(names) if k.startswith(self.t5_model_prefix)]]
        print("layer-wise")
        return [[i] for i in range(len(names))]

    def _accumulate(self, params):
        """Per-element accumulators for one round; (None, 1) when the score needs none."""
        return None, 1

    # ------------------------------------------------------------------ iteration (:162-207)
    def global_iterative_pruning(self, target_sparsity, dict_layers_to_prune, iteratation=1,
                                 max_sparsity_per_layer=1.0):
        kernels = self._kernels()
        # get_mask's protection step (:120-127) applies to the global / per-sub-model thresholds,
        # not to the layer-wise one (get_layerwise_mask, :144-157, has none)
        layerwise = not getattr(self, "is_global", False)
        names = [k for k, _ in self.model.named_parameters() if k in dict_layers_to_prune]
        params = [v for k, v in self.model.named_parameters() if k in dict_layers_to_prune]
        masks = [torch.ones(p.shape, dtype=torch.uint8, device=p.device) for p in params]
        t0 = time.time()
        for i in range(1, iteratation + 1):
            p_i = target_sparsity ** (iteratation / i)                                # (:166)
            accs, n_batches = self._accumulate(params)
            for group in self._threshold_groups(names):
                if not group:
                    raise RuntimeError("torch.cat of an empty score list (:130)")
                total = sum(params[j].numel() for j in group)
                k = int(p_i * total)                                                   # (:133)
                if k < 1:
                    raise IndexError("index -1 is out of bounds for dimension 0 with size 0")
                protect = None
                if not layerwise and max_sparsity_per_layer != 1.0:
                    protect = [int(params[j].numel() * (1 - max_sparsity_per_layer)) for j in group]
                kernels.global_threshold_prune(
                    [params[j].data for j in group],
                    None if accs is None else [accs[j] for j in group],
                    [masks[j] for j in group], self.score_mode, n_batches, k, protect)
            del accs
            print(f"Step {i}, target sparsity: {p_i:.4f}")
        self.stage_stats["global"] = {"seconds": time.time() - t0, "layers": len(names),
                                      "iterations": iteratation}
        return self.model

    @print_time
    def prune(self, importance_scores=None, keep_indices_or_masks=None):
        print("In: ", self.pruner_name)
        dtype_record, requires_grad_record, device = self.model_setup_and_record_attributes(self.model)
        if self.t5_prune_spec is None or self.vit_prune_spec is None:
            return self.model, None
        _, vit_keep_ratio, _, _ = self.convert_spec_to_list(self.vit_prune_spec)
        _, t5_keep_ratio, _, _ = self.convert_spec_to_list(self.t5_prune_spec)
        assert vit_keep_ratio == t5_keep_ratio

        def check(name, v):                                                            # (:221-228)
            return (len(v.shape) == 2 and ".block" in name
                    and "relative_attention_bias.weight" not in name
                    and (name.startswith(self.t5_model_prefix)
                         or name.startswith(self.vit_model_prefix)))

        parameters_to_prune = {k: v for k, v in self.model.named_parameters() if check(k, v)}
        self.model = self.global_iterative_pruning(
            1 - vit_keep_ratio, parameters_to_prune, iteratation=self.iteration,
            max_sparsity_per_layer=1.0)
        self.model_reset(self.model, dtype_record, requires_grad_record, device)
        return self.model, None


@registry.register_pruner("blipt5_global_mag_pruner")
class BLIPT5GlobalMagPruner(BLIPT5GlobalPruner):
    pruner_name = "blipt5_global_mag_pruner"
    score_mode = 3               # `v.data.float()` (:251)


@registry.register_pruner("blipt5_global_gradmagabs_pruner")
class BLIPT5GlobalGradMagAbsPruner(BLIPT5GlobalPruner):
    pruner_name = "blipt5_global_gradmagabs_pruner"
    score_mode = 0               # |W| * |mean_b |g||  (:298)

    def _accumulate(self, params):
        """(:259-296) — the accumulators stay in HBM, one multi-tensor launch per batch; data
        parallel as `LayerSparsity.accumulate_abs_grads` (one all-reduce per round)."""
        ls = getattr(self, "_grad_engine", None)
        if ls is None:            # one engine for all rounds: its captured fwd+bwd graph is reused
            ls = self._grad_engine = LayerSparsity(
                self.model, self.data_loader, loss_vision_language, self.num_samples, 0.0, 1.0,
                "GradMagAbs_sum", 1, 1e-3, {}, kernels=self._kernels(),
                process_group=self.process_group)
        return ls.accumulate_abs_grads(params)


@registry.register_pruner("blipt5_global_mezo_pruner")
class BLIPT5GlobalMeZoPruner(BLIPT5GlobalPruner):
    pruner_name = "blipt5_global_mezo_pruner"

    def global_iterative_pruning(self, target_sparsity, dict_layers_to_prune, iteratation=1,
                                 max_sparsity_per_layer=1.0):
        """One zeroth-order score per matrix (:323-389, the loop of a-3 with eps fixed at 1e-3
        and `self.num_samples`), so get_mask ranks matrices and drops whole ones."""
        names = [k for k, _ in self.model.named_parameters() if k in dict_layers_to_prune]
        layerwise = not getattr(self, "is_global", False)
        params = [v for k, v in self.model.named_parameters() if k in dict_layers_to_prune]
        mapping = {k: k for k in names}
        masks = None
        t0 = time.time()
        for i in range(1, iteratation + 1):
            p_i = target_sparsity ** (iteratation / i)
            ls = LayerSparsity(
                self.model, self.data_loader, loss_vision_language, self.num_samples,
                target_sparsity, 1.0, "MEZO-GradOnly_sum", self.num_noise, 1e-3, mapping,
                kernels=self.kernels, z_source=self.z_source, process_group=self.process_group)
            self.kernels = ls.kernels
            scores = ls.compute_importance_scores_mezo(mapping)
            scores = {k: scores[k].clone() for k in names}
            if masks is not None:
                for k in scores:
                    scores[k] *= masks[k]
            masks = {}
            for group in self._threshold_groups(names):
                sub = {names[j]: scores[names[j]] for j in group}
                if not layerwise:                      # get_mask's protection step (:120-127) on
                    for k_, v in sub.items():          # one-element score tensors
                        num_to_set = int(v.numel() * (1 - max_sparsity_per_layer))
                        if num_to_set > 0:
                            thr = torch.topk(v.flatten(), num_to_set, largest=True)[0][-1]
                            v[torch.where(v >= thr)] = torch.finfo(v.dtype).max
                all_scores = torch.cat([t.flatten() for t in sub.values()])
                num_to_zero_out = int(p_i * all_scores.numel())
                threshold = torch.topk(all_scores, num_to_zero_out, largest=False)[0][-1]
                for k, v in sub.items():
                    masks[k] = (v > threshold).type(v.dtype)
            for k, p in zip(names, params):
                if float(masks[k]) == 0.0:       # `v.data *= 0` (:200): every element -> +-0
                    self.kernels.mask_mul(p.data, torch.zeros(p.shape, dtype=torch.uint8,
                                                              device=p.device))
            print(f"Step {i}, target sparsity: {p_i:.4f}")
        self.stage_stats["global"] = {"seconds": time.time() - t0, "layers": len(names),
                                      "iterations": iteratation}
        return self.model
