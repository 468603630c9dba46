"""`BasePruner` / `LayerWiseBasePruner`: configuration holders of the pruner API
(LAVIS/lavis/compression/pruners/base_pruner.py:18-92 and
layer_single_base_pruner.py:19-117)."""
import time


import contextlib
import gc


@contextlib.contextmanager
def capture_graph(graph, **kw):
    """`torch.cuda.graph(graph, **kw)` with Python's cyclic garbage collector held off for the
    duration of the capture.  A collection that happens to run inside a capture may finalise
    objects that own device resources (an older loss closure's HIP graphs, pools, events); their
    destructors call into HIP on the capturing thread, which is illegal under
    capture_error_mode="thread_local" and aborts the process (seen once in ~3 runs of the full-size
    batched-evaluation test).  No collection is forced here (a full one costs ~85 ms on the
    BLIP-2 object graph and a stage-1 run captures hundreds of graphs): garbage simply waits
    until the capture has ended."""
    import torch
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was_enabled:
            gc.enable()


def print_time(func):
    """Wall-clock print around a stage, as the reference's decorator (pruners/utils.py:6-18)."""
    def wrapper(*args, **kwargs):
        start = time.time()
        ret = func(*args, **kwargs)
        print(f"{func.__name__} spent {time.time() - start:.3f} s")
        return ret
    wrapper.__name__ = func.__name__
    return wrapper


class BasePruner:
    def __init__(self, model, data_loader, is_strct_pruning, keep_indices_or_masks_cache,
                 importance_scores_cache, is_global, num_samples):
        self.model = model
        self.data_loader = data_loader
        self.is_strct_pruning = is_strct_pruning
        self.is_global = is_global
        self.num_samples = num_samples
        self.keep_indices_or_masks_cache = keep_indices_or_masks_cache
        self.importance_scores_cache = importance_scores_cache

    def prune(self, importance_scores=None, keep_indices_or_masks=None):
        raise NotImplementedError


class LayerWiseBasePruner(BasePruner):
    def __init__(self, model, data_loader, prune_spec=None, importance_scores_cache=None,
                 keep_indices_or_masks_cache=None, is_strct_pruning=False, num_samples=64,
                 is_global=False, model_prefix="t5_model", sparsity_ratio_granularity=None,
                 max_sparsity_per_layer=0.8, score_method="GradMagSquare_avg",
                 num_data_first_stage=128, num_noise=1, sparsity_dict=None, noise_eps=1e-3,
                 prune_per_model=False, kernels=None, z_source="torch", process_group=None,
                 prefix_cache=True, use_graphs=True, n_lanes=2, eval_batch=0, k1_form="block",
                 k6_immediate=False, stage1_checkpoint=None, **kwargs):
        super().__init__(model=model, data_loader=data_loader, is_strct_pruning=is_strct_pruning,
                         importance_scores_cache=importance_scores_cache,
                         keep_indices_or_masks_cache=keep_indices_or_masks_cache,
                         is_global=is_global, num_samples=num_samples)
        self.sparsity_ratio_granularity = sparsity_ratio_granularity
        self.max_sparsity_per_layer = max_sparsity_per_layer
        self.score_method = score_method
        self.num_data_first_stage = num_data_first_stage
        self.num_noise = num_noise
        self.sparsity_dict = sparsity_dict
        self.noise_eps = noise_eps
        self.prune_per_model = prune_per_model
        self.prune_spec = prune_spec
        self.model_prefix = model_prefix
        self.prune_n = 0
        self.prune_m = 0
        self.model_stem = getattr(self.model, model_prefix, None)
        # build-side extras (keyword only in the reference's **kwargs slot)
        self.kernels = kernels
        self.z_source = z_source
        self.process_group = process_group
        self.prefix_cache = prefix_cache
        self.use_graphs = use_graphs
        self.n_lanes = n_lanes
        # loss evaluations of a layer per pass of the shared suffix; 0: sized from the calibration
        # set where stage 1 starts (pruners/wanda.py::_eval_batch)
        self.eval_batch = eval_batch
        self.k1_form = k1_form
        # stage 2: reduce each hooked Linear input inside its hook (one launch per input, as the
        # reference does) instead of one launch per block — for models that write their
        # activations in place after the Linear ran (pruners/wanda.py::_K6Collector.add)
        self.k6_immediate = k6_immediate
        # zeroth-order stage 1: path of a resumable checkpoint of the loss table (LayerSparsity's
        # `checkpoint_path`)
        self.stage1_checkpoint = stage1_checkpoint
        self.stage_stats = {}

    def model_setup_and_record_attributes(self, model):
        """Record dtypes / requires_grad and enable grads on everything (:79-95)."""
        dtype_record = {n: p.data.dtype for n, p in model.named_parameters()}
        requires_grad_record = {}
        for n, p in model.named_parameters():
            requires_grad_record[n] = p.requires_grad
            p.requires_grad = True
        device = next(iter(model.parameters())).device
        if device.type == "cuda" and hasattr(model, "stage_plan"):
            # the build's own shape modules: their 16-bit Linears run ONE pinned hipBLASLt solution
            # per weight shape (shapes/fused.py; a model built on the CPU and moved over gets it here)
            from ..shapes.fused import pin_linears
            pin_linears(model)
        if device.type == "cuda" and not hasattr(model, "stage_plan"):
            # a model the caller owns: its convolutions are MIOpen's on this platform, and MIOpen
            # picks their kernel by TIMING candidates on first use (round 6: three roundings of one
            # patch embedding, by process and by what ran on the box before) — said once, up front
            from ..shapes.fused import unpinned_convs
            convs = unpinned_convs(model)
            if convs:
                import warnings
                warnings.warn(
                    f"{len(convs)} convolution(s) ({', '.join(convs[:3])}{', …' if len(convs) > 3 else ''}) "
                    "run through MIOpen, which selects their kernel by timing: two runs (or two "
                    "data-parallel ranks) may round them differently and end with different "
                    "tables.  ecoflap_amd.shapes.fused.pin_patch_convs(model) runs a patch "
                    "embedding as a GEMM instead; MIOPEN_DEBUG_FIND_ONLY_SOLVER pins MIOpen's choice.",
                    RuntimeWarning, stacklevel=3)
        if device.type == "cuda":
            # which GEMM each weight shape runs is bound afresh for this run, as in a fresh process
            # (shapes/fused.py: begin_run) — not inherited from whatever the process ran before
            from ..shapes import fused
            fused.begin_run()
            from .. import blas_guard
            # the GEMM library must be in its reproducible mode; batch invariance on top only
            # when evaluations are going to be concatenated
            blas_guard.verify(device, need_batch_invariance=int(getattr(self, "eval_batch", 1)) != 1)
        return dtype_record, requires_grad_record, device

    def model_reset(self, model, dtype_record, requires_grad_record, device):
        for n, p in model.named_parameters():
            p.requires_grad = requires_grad_record[n]
        for n, p in model.named_parameters():
            if p.data.dtype != dtype_record[n]:
                p.data = p.data.type(dtype_record[n])
        model.to(device)

    def convert_spec_to_list(self, spec):
        """"<n_layers>-<keep>-<attn>-<ffn>" (:108-114)."""
        num_layers, res_keep, attn_keep, ffn_keep = spec.split("-")
        return int(num_layers), float(res_keep), float(attn_keep), float(ffn_keep)
