"""Stage 2 of ECoFLaP: Wanda local pruning, block by block, driven by the stage-1
sparsity table; plus the three registered pruners of the reference.

Host-side mirror of LAVIS/lavis/compression/pruners/wanda_pruner.py:
  WrappedGPT (:54-84)                      -> K6 `ecoflap_colsqnorm_accum`
  T5LayerWandaPruner (:87-375)             -> rows mode of K7 (:272-279)
  VITLayerWandaPruner (:378-657)           -> matrix mode of K7 (:555-558)
  BLIPT5LayerWandaPruner (:660-875)        -> ViT blocks, T5 encoder, T5 decoder
Same registered names, constructor keywords, `prune()` contract
(`-> (model, sparsity_dict_or_None)`, weights pruned in place) and sparsity-table
keys.  The column statistics, the metric, the k-smallest selection and the zeroing
run as HIP kernels on the weight's own storage; nothing is sorted and no
metric/mask tensor is materialised.
"""
import torch
import torch.nn as nn

from .. import hip as _hip
from ..registry import registry
from .base_pruner import LayerWiseBasePruner, capture_graph, print_time
from .layer_sparsity import LayerSparsity, _default_batch_len
from .losses import loss_language, loss_vision, loss_vision_language


def get_module_recursive(base, module_to_process):
    for part in [p for p in module_to_process.split(".") if p]:
        base = getattr(base, part)
    return base


def find_layers(module, layers=(nn.Linear,), name=""):
    """name -> module for every nn.Linear below `module` (wanda_pruner.py:33-52)."""
    if type(module) in tuple(layers):
        return {name: module}
    res = {}
    for child_name, child in module.named_children():
        res.update(find_layers(child, layers=layers,
                               name=name + "." + child_name if name != "" else child_name))
    return res


class WrappedGPT:
    """Running mean of the per-input-channel sum of squares of a Linear's inputs."""

    def __init__(self, layer, layer_id=0, layer_name="none", kernels=None):
        self.layer = layer
        self.dev = self.layer.weight.device
        self.rows = layer.weight.data.shape[0]
        self.columns = layer.weight.data.shape[1]
        self.scaler_row = torch.zeros((self.columns), device=self.dev)
        self.nsamples = 0
        self.layer_id = layer_id
        self.layer_name = layer_name
        self.kernels = kernels if kernels is not None else _hip.HipKernels()
        self.n_dev = None          # device-side sample count while a block graph is captured / replayed
        self.dev_ws = None
        self.dev_batch = 0
        self.sink = None           # block-level collector: ONE K6 launch per block and sample

    def add_batch(self, inp, out):
        if len(inp.shape) == 2:
            inp = inp.unsqueeze(0)
        tmp = inp.shape[0]
        x = inp.reshape((-1, inp.shape[-1]))
        if not x.is_contiguous():
            x = x.contiguous()
        if self.sink is not None:
            self.sink.add(self, x, tmp)    # launched with the block's other inputs (`flush`)
            return
        if self.n_dev is not None:
            # graph capture: same update, the count lives on the device (the replay loop keeps
            # `nsamples` in step on the host)
            if self.dev_ws is None:
                self.dev_ws = self.kernels.colsqnorm_workspace(x.shape[0], x.shape[1], x.device)
            self.kernels.colsqnorm_accum_dev(self.scaler_row, x, self.n_dev, tmp, self.dev_ws)
            self.dev_batch = tmp
            return
        self.kernels.colsqnorm_accum(self.scaler_row, x, self.nsamples, tmp)
        self.nsamples += tmp


class _K6Collector:
    """The forward hooks of a block's Linears hand their inputs over here; `flush()` (after the
    block's forward, also inside a graph capture) reduces all of them in ONE launch
    (`ecoflap_colsqnorm_accum_multi`).  Same arithmetic per input as one `add_batch` each.

    raw=True (data-parallel stage 2): each input's own statistic ||x_c||^2 goes into a flat
    per-sample row instead of the running mean; `merge()` exchanges the rows of all ranks'
    batches and replays the mean in GLOBAL batch order (`ecoflap_colsq_replay`), so every rank
    ends with the one-process scaler_row bit for bit."""

    def __init__(self, kernels, raw=False, immediate=False):
        self.kernels, self.raw = kernels, raw
        self.immediate = bool(immediate)    # reduce inside the hook (models with in-place ops)
        self.pending = []
        self._flushes = []         # item lists of the current sample's launches
        self._sample_flushes = []  # ... of the last finished sample (sizes the private workspace)
        self.ws = None             # private workspace while a graph is captured / replayed
        self.sites = None          # raw: [(wrapped, offset, cols)] in hook order (one forward)
        self.flat = None           # raw: static [sum cols] target of one forward
        self.site_batches = None   # raw: batch (leading dim) per site
        self.samples = []          # raw: one flat row per local calibration sample
        self.launches = 0
        self.capturing = False     # inside a graph capture: launches are recorded, not run

    def add(self, wrapped, x, batch):
        """`x` is kept BY REFERENCE until `flush()` (normally a view of the activation the hook
        saw): the block must not write its Linear inputs in place after the Linear ran (in-place
        residual adds / activations on that tensor, fused ops writing into their input).  The
        reference reduces inside the hook (wanda_pruner.py:71-84) and has no such requirement, so
        it is checked: the tensor's version counter is recorded here and compared at flush."""
        if any(p[0] is wrapped for p in self.pending):
            self.flush()           # a Linear called twice in one forward: in order, not at once
        if self.immediate:
            self.pending.append((wrapped, x, batch, x._version))
            self.flush()
            return
        self.pending.append((wrapped, x, batch, x._version))

    def flush(self):
        if not self.pending:
            return
        pend, self.pending = self.pending, []
        stale = [w_ for w_, x, _, ver in pend if x._version != ver]
        if stale:
            raise RuntimeError(
                f"{len(stale)} hooked Linear input(s) of this block were modified in place between "
                "the Linear's forward and the end of the block: the deferred one-launch column "
                "statistic would read the overwritten values.  Construct the pruner with "
                "k6_immediate=True (one launch per input, inside the hook, as the reference "
                "reduces) for models that write their activations in place.")
        pend = [(w_, x, b) for w_, x, b, _ in pend]
        if self.raw:
            # one forward may flush more than once (a Linear called twice): sites accumulate
            # until `end_sample`
            if self.flat is None or self._site_cursor + len(pend) > len(self.sites or []):
                self._grow_sites(pend)
            items = []
            for w_, x, b in pend:
                w0, off, cols = self.sites[self._site_cursor]
                assert w0 is w_ and cols == x.shape[1]
                items.append((self.flat[off:off + cols], x, 0, None, b, True))
                self.site_batches[self._site_cursor] = b
                self._site_cursor += 1
                if not self.capturing:
                    w_.nsamples += b
        else:
            items = [(w_.scaler_row, x, w_.nsamples, w_.n_dev, b, False) for w_, x, b in pend]
            for w_, _, b in pend:
                if w_.n_dev is None:
                    w_.nsamples += b
                else:
                    w_.dev_batch = b
        self.kernels.colsqnorm_accum_multi(items, self.ws)
        self.launches += 1
        self._flushes.append(items)

    _site_cursor = 0

    def _grow_sites(self, pend):
        assert not self.samples, "hook sequence changed between calibration samples"
        sites = list(self.sites or [])
        off = sum(c for _, _, c in sites)
        for w_, x, _ in pend:
            sites.append((w_, off, x.shape[1]))
            off += x.shape[1]
        old = self.flat
        self.flat = torch.zeros(off, dtype=torch.float32, device=pend[0][1].device)
        if old is not None:
            self.flat[:old.numel()].copy_(old)
        self.sites = sites
        self.site_batches = (self.site_batches or []) + [0] * len(pend)

    def end_sample(self, replayed=False):
        """after one calibration sample's forward (+ flush), run eagerly or replayed from the
        captured graph (`replayed`: the host-side counts did not move with the launch)"""
        if self.raw:
            if not replayed:
                assert self._site_cursor == len(self.sites)
            else:
                for (w_, _, _), b in zip(self.sites, self.site_batches):
                    w_.nsamples += b
            self.samples.append((self.flat.clone(), list(self.site_batches)))
        self._site_cursor = 0
        if self._flushes:
            self._sample_flushes, self._flushes = self._flushes, []

    def private_workspace(self):
        """for graph capture: a workspace of its own for the launches of the sample just run
        eagerly — sized for the LARGEST of them (a Linear called twice per forward flushes more
        than once; the launches of one graph run one after the other and the tickets in the
        workspace reset themselves, so they can share it)"""
        sizes = [self.kernels.colsqnorm_multi_workspace(items) for items in self._sample_flushes]
        if not sizes or any(w is None for w in sizes):
            self.ws = None
        else:
            self.ws = max(sizes, key=lambda w: w.numel())
        return self.ws is not None

    def merge(self, local_batch_ids, n_global, process_group):
        """raw mode, after the block's pass: all ranks' per-batch rows -> every rank replays the
        running mean over the global batch order.  ONE all-reduce of a [batches, sum cols] fp32
        matrix whose rows each have exactly one non-zero contributor (exact)."""
        import torch.distributed as dist
        total = self.flat.numel()
        dev = self.flat.device
        glob = torch.zeros(n_global, total + len(self.sites), dtype=torch.float32, device=dev)
        for bi, (row, bs) in zip(local_batch_ids, self.samples):
            glob[bi, :total] = row
            glob[bi, total:] = torch.tensor(bs, dtype=torch.float32, device=dev)   # small ints: exact
        dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=process_group)
        batches = glob[:, total:].round().to(torch.int64).cpu().tolist()
        by_w = {}
        for si, (w_, off, cols) in enumerate(self.sites):
            by_w.setdefault(id(w_), (w_, []))[1].append((si, off, cols))
        for w_, sites in by_w.values():
            if len(sites) == 1:
                si, off, cols = sites[0]
                sq = glob[:, off:off + cols]
                bl = [batches[j][si] for j in range(n_global)]
            else:        # called several times per forward: (sample, site) order
                sq = torch.stack([glob[j, off:off + cols] for j in range(n_global)
                                  for _, off, cols in sites]).contiguous()
                bl = [batches[j][si] for j in range(n_global) for si, _, _ in sites]
            w_.scaler_row.zero_()
            self.kernels.colsq_replay(w_.scaler_row, sq, bl, 0)
            w_.nsamples_global = int(sum(bl))


class _StopForward(Exception):
    """Raised from the block-0 pre-hook once its inputs are recorded."""


T5_BLOCK_KWARGS = [
    "attention_mask", "position_bias", "encoder_attention_mask", "encoder_decoder_position_bias",
    "layer_head_mask", "cross_attn_layer_head_mask", "encoder_hidden_states",
]


class _BlockwiseWanda:
    """Calibration capture + per-block statistics + selection, shared by the pruners."""

    def __init__(self, owner):
        self.owner = owner
        self.kernels = owner.kernels if owner.kernels is not None else _hip.HipKernels()
        owner.kernels = self.kernels

    def _rank_world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            g = getattr(self.owner, "process_group", None)
            return dist.get_rank(g), dist.get_world_size(g)
        return 0, 1

    def _merge_statistics(self, wrapped):
        """Data-parallel stage 2: ONE all-reduce per block of the concatenated per-channel
        sums (running mean x local sample count) and the sample counts; every rank then holds
        bit-identical statistics and computes identical masks."""
        rank, world = self._rank_world()
        if world == 1:
            return
        import torch.distributed as dist
        names = list(wrapped)
        dev = wrapped[names[0]].scaler_row.device
        parts = [wrapped[n].scaler_row.double() * wrapped[n].nsamples for n in names]
        counts = torch.tensor([float(wrapped[n].nsamples) for n in names], dtype=torch.float64,
                              device=dev)
        flat = torch.cat(parts + [counts])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=getattr(self.owner, "process_group", None))
        off = 0
        totals = flat[-len(names):]
        for i, n in enumerate(names):
            k = wrapped[n].scaler_row.numel()
            wrapped[n].scaler_row.copy_((flat[off:off + k] / totals[i]).float())
            wrapped[n].nsamples_global = int(totals[i].item())
            off += k

    def _merge_hessians(self, wrapped):
        """Data-parallel SparseGPT: each rank's H is (2/n_r) * sum_r x x^T over its own calibration
        batches; the global Hessian is sum_r (n_r / N) H_r.  ONE all-reduce per block of the
        concatenated n_r * H_r (fp32; 175 MB for a ViT-g block, 0.6 GB for a T5 decoder block —
        a ring over xGMI) plus the counts; every rank then runs the same OBS sweeps on the same
        bits and replicas stay identical.  Sums re-associate across ranks, so the result equals
        the single-process run to rounding, not bit for bit."""
        rank, world = self._rank_world()
        if world == 1:
            return
        import torch.distributed as dist
        names = list(wrapped)
        dev = wrapped[names[0]].H.device
        flat = torch.cat([(wrapped[n].H * float(wrapped[n].nsamples)).reshape(-1) for n in names]
                         + [torch.tensor([float(wrapped[n].nsamples) for n in names],
                                         dtype=torch.float32, device=dev)])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=getattr(self.owner, "process_group", None))
        totals = flat[-len(names):]
        off = 0
        for i, n in enumerate(names):
            k = wrapped[n].H.numel()
            wrapped[n].H.copy_((flat[off:off + k] / totals[i]).view_as(wrapped[n].H))
            wrapped[n].nsamples_global = int(totals[i].item())
            off += k

    def capture(self, model, dataloader, blocks, forward_fn, cache_keys, n_samples,
                optional_keys=False, batch_len=None):
        """Record the inputs of block 0 for the first n_samples calibration samples
        (the reference swaps block 0 for a `Catcher`, :184-209 / :469-493; a pre-hook
        leaves the ModuleList untouched)."""
        inps, caches = [], []

        def grab(_module, args, kwargs):
            x = args[0]
            inps.append(x.detach())
            cache = {}
            for key in cache_keys:
                if key in kwargs:
                    cache[key] = kwargs[key]
                elif len(cache_keys) == 1 and len(args) > 1:
                    cache[key] = args[1]          # ViT: blk(x, rel_pos_bias) positional
                elif optional_keys:
                    continue                      # UPop's BERT Catcher keeps the keys present
                else:
                    raise KeyError(key)           # as the reference's Catcher would
            caches.append(cache)
            raise _StopForward()

        handle = blocks[0].register_forward_pre_hook(grab, with_kwargs=True)
        total = 0
        rank, world = self._rank_world()
        self.local_batch_ids, self.global_batches = [], 0
        try:
            for bi, batch in enumerate(dataloader):
                if total >= n_samples:
                    break
                self.global_batches = bi + 1
                if batch_len is not None:
                    total += batch_len(batch)
                else:
                    total += (_default_batch_len(batch) if "image" not in batch
                              else batch["image"].shape[0])
                if (bi % world) != rank:
                    continue              # another rank calibrates on this batch
                self.local_batch_ids.append(bi)
                try:
                    forward_fn(model, batch)
                except _StopForward:
                    pass
        finally:
            handle.remove()
        return inps, [None] * len(inps), caches

    def run(self, model, dataloader, module_to_process, n_samples, sparsity_ratio, forward_fn,
            cache_keys, autocast, take_first, mode, optional_keys=False, batch_len=None,
            count_factor=1):
        import time
        t0 = time.time()
        try:
            return self._run(model, dataloader, module_to_process, n_samples, sparsity_ratio,
                             forward_fn, cache_keys, autocast, take_first, mode, optional_keys,
                             batch_len, count_factor)
        finally:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self.owner.stage_stats.setdefault("stage2", {})[module_to_process] = time.time() - t0

    def _run(self, model, dataloader, module_to_process, n_samples, sparsity_ratio, forward_fn,
             cache_keys, autocast, take_first, mode, optional_keys=False, batch_len=None,
             count_factor=1):
        from .phase_timer import PhaseTimer
        with torch.no_grad(), PhaseTimer.span("stage2.capture (the model's forward up to block 0, per sample)"):
            blocks = get_module_recursive(model, module_to_process)
            inps, outs, caches = self.capture(model, dataloader, blocks, forward_fn, cache_keys,
                                              n_samples, optional_keys, batch_len)
        n_batches = min(n_samples, len(inps))     # (:226/:505: compared against the batch count)

        collector = [None]        # the live block's K6 collector (None: plain pass / SparseGPT)

        def call(block, j):
            with torch.no_grad(), autocast():
                y = block(inps[j], **caches[j])
            if collector[0] is not None:
                collector[0].flush()
                collector[0].end_sample()
            return y[0] if take_first else y

        # equally shaped calibration samples on the GPU: each block forward (with its K6 hooks
        # in the first pass) is captured once as a HIP graph and replayed per sample — the loop
        # is launch-bound (the reference's batch 1: ~40 kernels of a few microseconds per block)
        graphed_plain = (torch.is_tensor(inps[0]) and inps[0].is_cuda
                         and n_batches >= int(getattr(self.owner, "graph_min_batches", 24))   # two captures per block cost ~5 ms
                         and bool(getattr(self.owner, "use_graphs", True))
                         and bool(getattr(self.kernels, "graph_safe", False))
                         and _uniform_calibration(inps, caches, n_batches))
        # SparseGPT's hooks keep their inputs by reference until 8 samples are in (one MFMA call):
        # a replayed graph would overwrite them, so its hooked pass stays eager; the pass behind
        # the pruning has no hooks and replays like Wanda's (round 5: 3.9 s of 128-sample eager
        # block forwards at batch 1 in the BLIP-2 run)
        graphed = graphed_plain and getattr(self.owner, "local_method", "wanda") != "sparsegpt"
        # Batch-1 calibration samples (the launchers' default) leave a replayed block forward at
        # ~40 kernels of a few microseconds: G samples ride one replay where the block gives
        # every slot of the stacked input the bits it gives that sample alone (checked on the
        # first group of every block: slots 0 and G-1 against their batch-1 forwards; a part
        # whose check fails once goes back to one replay per sample).  round 6: 7.1 s of
        # SparseGPT's 15 s of stage 2 on the BLIP-2 shape were these forwards.
        G = int(getattr(self.owner, "stage2_group", 8))
        min_values = int(getattr(self.owner, "stage2_group_min_values", 65536))
        group_ok = [graphed_plain and G > 1 and int(inps[0].shape[0]) == 1 and n_batches >= 2 * G
                    and all((not torch.is_tensor(v)) or (v.dim() > 0 and int(v.shape[0]) == 1)
                            for v in caches[0].values())]

        def stacked_inputs():
            sx = torch.cat([inps[j] for j in range(G)], 0)
            skw = {k: (torch.cat([caches[j][k] for j in range(G)], 0) if torch.is_tensor(v) else v)
                   for k, v in caches[0].items()}
            return sx, skw

        def load_group(sx, skw, g):
            from ..shapes import fused
            pairs = []
            for i in range(G):
                j = g * G + i
                pairs.append((sx[i:i + 1], inps[j]))
                pairs += [(skw[k][i:i + 1], v) for k, v in caches[j].items() if torch.is_tensor(v)]
            fused.multi_copy(pairs)

        def graph_pass_grouped(block):
            """The plain pass (no hooks) with G samples per replay -> False: not slot invariant
            here (nothing kept; the caller runs `graph_pass`)."""
            y_first, y_last = call(block, 0), call(block, G - 1)
            sx, skw = stacked_inputs()
            with torch.no_grad(), autocast():
                block(sx, **skw)                                  # warm-up at this width
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), capture_graph(graph, capture_error_mode="thread_local"):
                with autocast():
                    y = block(sx, **skw)
                y = y[0] if take_first else y
            graph.replay()
            if not (torch.equal(y[0:1], y_first) and torch.equal(y[G - 1:G], y_last)):
                torch.cuda.current_stream().synchronize()
                del graph
                return False
            # too few values for two slots to rule out a lucky agreement (toy shapes: a handful
            # of short dot products round alike in two kernels most of the time): every sample
            # is then also run alone and compared, and the first difference ends the grouping
            verify_all = 2 * y_first.numel() < min_values
            n_full = n_batches // G
            rest_from = n_full * G
            for g in range(n_full):
                if g:
                    load_group(sx, skw, g)
                    graph.replay()
                yc = y.clone()
                if verify_all:
                    alone = [call(block, g * G + i) for i in range(G)]
                    if not all(torch.equal(yc[i:i + 1], alone[i]) for i in range(G)):
                        for i in range(G):
                            outs[g * G + i] = alone[i]
                        rest_from, group_ok[0] = (g + 1) * G, False
                        break
                for i in range(G):
                    outs[g * G + i] = yc[i:i + 1]
            for j in range(rest_from, n_batches):
                outs[j] = call(block, j)
            torch.cuda.current_stream().synchronize()     # the graph's buffers go away with it
            del graph
            self.owner.stage_stats["stage2_grouped_passes"] = (
                self.owner.stage_stats.get("stage2_grouped_passes", 0) + 1)
            return True

        def graph_pass_hessian_grouped(block, wrapped):
            """SparseGPT's hooked pass with G = `samples_per_call` samples per replay: the static
            input each Linear read IS the concatenated [G x tokens, cols] operand of one
            `hessian_accum` call — the calls, their rows and their order are those of
            `graph_pass_hessian`.  Checked per block: every Linear's rows of slot 0 and of slot
            G-1 against the inputs it sees in batch-1 forwards of samples 0 and G-1.
            -> False (nothing accumulated) when they differ or a Linear is off the MFMA path."""
            from .sparsegpt import SparseGPT
            if G != int(SparseGPT.samples_per_call):
                return False
            seen = {}

            def noting(name):
                def add_batch(inp, out):
                    seen[name] = inp.reshape((-1, inp.shape[-1])).clone(memory_format=torch.contiguous_format)
                return add_batch

            if any(w_.nsamples != 0 or w_._pending for w_ in wrapped.values()):
                return False
            for name, w_ in wrapped.items():
                w_.add_batch = noting(name)
            try:
                call(block, 0)
                ref_first, seen = seen, {}
                call(block, G - 1)
                ref_last, seen = seen, {}
                sx, skw = stacked_inputs()
                with torch.no_grad(), autocast():
                    block(sx, **skw)                              # warm-up at this width
                seen = {}
                graph = torch.cuda.CUDAGraph()
                with torch.no_grad(), capture_graph(graph, capture_error_mode="thread_local"):
                    with autocast():
                        block(sx, **skw)
                static, seen = seen, {}
                graph.replay()
                ok = set(static) == set(wrapped) == set(ref_first)
                for name in (static if ok else ()):
                    t = ref_first[name].shape[0]
                    ok = (ok and static[name].shape[0] == G * t
                          and static[name].dtype in (torch.float16, torch.bfloat16)
                          and torch.equal(static[name][:t], ref_first[name])
                          and torch.equal(static[name][(G - 1) * t:], ref_last[name]))
                if not ok:
                    torch.cuda.current_stream().synchronize()
                    del graph
                    return False
                verify_all = 2 * sum(v.numel() for v in ref_first.values()) < min_values   # (see graph_pass_grouped)
                n_full = n_batches // G
                rest_from = n_full * G
                for g in range(n_full):
                    if g:
                        load_group(sx, skw, g)
                        graph.replay()
                    if verify_all:
                        same = True
                        for i in range(G):
                            seen = {}
                            call(block, g * G + i)               # the noting hooks: this sample alone
                            same = same and all(
                                torch.equal(static[n][i * seen[n].shape[0]:(i + 1) * seen[n].shape[0]], seen[n])
                                for n in static)
                        if not same:
                            rest_from, group_ok[0] = g * G, False
                            break
                    for name, w_ in wrapped.items():
                        w_.kernels.hessian_accum(w_.H, static[name], w_.nsamples, G)
                        w_.nsamples += G
            finally:
                for w_ in wrapped.values():
                    del w_.add_batch
            for j in range(rest_from, n_batches):
                call(block, j)                                    # the ordinary hooks; flushed by the caller
            torch.cuda.current_stream().synchronize()
            del graph
            self.owner.stage_stats["stage2_grouped_passes"] = (
                self.owner.stage_stats.get("stage2_grouped_passes", 0) + 1)
            return True

        def graph_pass(block, wrapped, keep):
            """All n_batches samples through `block`: sample 0 eagerly (warm-up), one capture,
            n-1 replays.  wrapped: the K6 statistics whose hooks are live (None: plain pass)."""
            y0 = call(block, 0)
            if keep:
                outs[0] = y0
            static_x = inps[0].clone()
            static_kw = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in caches[0].items()}
            col = collector[0] if wrapped else None
            for w_ in (wrapped or {}).values():
                if col is None or not col.raw:
                    w_.n_dev = torch.tensor([w_.nsamples], dtype=torch.int64, device=static_x.device)
            if col is not None:
                col.private_workspace()       # (None: the cached one; this stream only)
                col.capturing = True
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), capture_graph(graph, capture_error_mode="thread_local"):
                with autocast():
                    y = block(static_x, **static_kw)
                if col is not None:
                    col.flush()
                y = y[0] if take_first else y
            if col is not None:
                col.capturing = False
                col._site_cursor = 0
            for j in range(1, n_batches):
                static_x.copy_(inps[j], non_blocking=True)
                for k, v in caches[j].items():
                    if torch.is_tensor(v):
                        static_kw[k].copy_(v, non_blocking=True)
                graph.replay()
                if keep:
                    outs[j] = y.clone()
                if col is not None and col.raw:
                    col.end_sample(replayed=True)
                else:
                    for w_ in (wrapped or {}).values():
                        w_.nsamples += w_.dev_batch
            for w_ in (wrapped or {}).values():
                w_.n_dev, w_.dev_ws = None, None
            if col is not None:
                col.ws = None
            torch.cuda.current_stream().synchronize()     # the graph's buffers go away with it
            del graph

        def graph_pass_hessian(block, wrapped):
            """SparseGPT's hooked pass as graph replays (round 5: 4.3 s of eager block forwards at
            batch 1 in the BLIP-2 run).  The reference's hook reduces each input into H at once
            (sparsegpt_pruner.py:71-82); this build's keeps `samples_per_call` inputs and makes ONE
            MFMA call over them, which a replayed graph would break (it overwrites the tensors
            the hooks kept by reference).  So: sample 0 runs eagerly with the ordinary hooks; the
            capture's hooks only note WHICH static tensors the Linears read; after every replay
            one fused copy moves those into slot s of a ring [S, tokens, cols] per Linear, and
            every S samples each Linear's ring — already the concatenated layout, no torch.cat —
            goes through `hessian_accum`.  Same inputs in the same order through the same kernel:
            the Hessians are those of the eager pass bit for bit.  -> False when a Linear is not on
            the MFMA path (fp32 activations): the caller runs the pass eagerly."""
            from ..shapes import fused
            from .sparsegpt import SparseGPT
            outs[0] = call(block, 0)
            S_ = int(SparseGPT.samples_per_call)
            first = {}
            for name, w_ in wrapped.items():
                if len(w_._pending) != 1 or w_.nsamples != 0:
                    return False
                first[name] = w_._pending[0]
            # (the eager path's guard, SparseGPT.flush: an input written in place after its hook saw
            # it is not the tensor the reference reduced — leave the pass to that path, whose flush
            # raises with the explanation)
            if any(x0._version != ver0 for x0, _, ver0 in first.values()):
                return False
            rings, static, slot = {}, {}, 1
            for name, (x0, b0, _) in first.items():
                rings[name] = torch.empty((S_,) + tuple(x0.shape), dtype=x0.dtype, device=x0.device)
                rings[name][0].copy_(x0)
                wrapped[name]._pending = []

            def flush():
                nonlocal slot
                for name, w_ in wrapped.items():
                    r = rings[name]
                    b_tot = first[name][1] * slot
                    w_.kernels.hessian_accum(w_.H, r[:slot].reshape(slot * r.shape[1], r.shape[2]),
                                             w_.nsamples, b_tot)
                    w_.nsamples += b_tot
                slot = 0

            def recorder(name):
                def add_batch(inp, out):
                    # a copy taken INSIDE the capture: the graph records the value at hook time, so
                    # a block that writes this input in place later in its forward cannot change
                    # what reaches the ring after the replay
                    static[name] = inp.reshape((-1, inp.shape[-1])).clone(memory_format=torch.contiguous_format)
                return add_batch

            static_x = inps[0].clone()
            static_kw = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in caches[0].items()}
            for name, w_ in wrapped.items():
                w_.add_batch = recorder(name)
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.no_grad(), capture_graph(graph, capture_error_mode="thread_local"):
                    with autocast():
                        block(static_x, **static_kw)
            finally:
                for w_ in wrapped.values():
                    del w_.add_batch
            assert set(static) == set(wrapped) and all(
                static[n].shape == rings[n].shape[1:] and static[n].dtype == rings[n].dtype for n in static)
            for j in range(1, n_batches):
                static_x.copy_(inps[j], non_blocking=True)
                for k, v in caches[j].items():
                    if torch.is_tensor(v):
                        static_kw[k].copy_(v, non_blocking=True)
                graph.replay()
                fused.multi_copy([(rings[n][slot], static[n]) for n in static])
                slot += 1
                if slot == S_:
                    flush()
            if slot:
                flush()
            torch.cuda.current_stream().synchronize()     # the graph's buffers go away with it
            del graph
            return True

        for i in range(len(blocks)):
            block = blocks[i]
            subset = find_layers(block)
            sparsegpt = getattr(self.owner, "local_method", "wanda") == "sparsegpt"
            if sparsegpt:
                from .sparsegpt import SparseGPT
                wrapped = {name: SparseGPT(subset[name], kernels=self.kernels) for name in subset}
            else:
                wrapped = {name: WrappedGPT(subset[name], kernels=self.kernels) for name in subset}
            handles = [
                subset[name].register_forward_hook(
                    lambda _m, inp, out, _n=name: wrapped[_n].add_batch(inp[0].data, out.data))
                for name in wrapped
            ]
            _, world = self._rank_world()
            if not sparsegpt and hasattr(self.kernels, "colsqnorm_accum_multi"):
                # ONE K6 launch per block and sample; under data parallelism the per-batch
                # statistics are kept apart and replayed in global order (exact)
                collector[0] = _K6Collector(
                    self.kernels, raw=world > 1 and hasattr(self.kernels, "colsq_replay"),
                    immediate=bool(getattr(self.owner, "k6_immediate", False)))
                for w_ in wrapped.values():
                    w_.sink = collector[0]
            from .phase_timer import PhaseTimer
            with PhaseTimer.span("stage2.block_forward_with_hooks (incl. K6 / Hessian updates)"):
                hess_graphs = sparsegpt and graphed_plain and getattr(self.owner, "sparsegpt_graph_hooks", True)
                grouped = False
                if hess_graphs and group_ok[0]:
                    grouped = graph_pass_hessian_grouped(block, wrapped)
                    group_ok[0] = grouped
                if graphed:
                    graph_pass(block, wrapped, keep=False)
                elif grouped:
                    pass
                elif not (hess_graphs and graph_pass_hessian(block, wrapped)):
                    # (a refused graph pass has run sample 0 already: its hooks have fired)
                    done0 = sparsegpt and any(w_.nsamples or w_._pending for w_ in wrapped.values())
                    for j in range(1 if done0 else 0, n_batches):
                        outs[j] = call(block, j)
            for h in handles:
                h.remove()
            if sparsegpt:
                for w_ in wrapped.values():
                    w_.flush()                 # buffered calibration samples -> H
            col, collector[0] = collector[0], None
            for w_ in wrapped.values():
                w_.sink = None
            if col is not None:
                self.owner.stage_stats.setdefault("k6_launches", 0)
                self.owner.stage_stats["k6_launches"] += col.launches
            if col is not None and col.raw:
                self.owner.stage_stats["k6_dp_exact_blocks"] = (
                    self.owner.stage_stats.get("k6_dp_exact_blocks", 0) + 1)
                col.merge(self.local_batch_ids[:n_batches], self.global_batches,
                          getattr(self.owner, "process_group", None))
            elif not sparsegpt:
                self._merge_statistics(wrapped)
            else:
                self._merge_hessians(wrapped)
            twins = {}
            if sparsegpt:      # Linears fed by the same tensor have bit-identical Hessians
                with PhaseTimer.span("sparsegpt.twin_detection (torch.equal of Hessians)"):
                    names_ = list(subset)
                    for a_i, a in enumerate(names_):
                        for b in names_[:a_i]:
                            if (b not in twins and wrapped[b].H.shape == wrapped[a].H.shape
                                    and torch.equal(wrapped[b].H, wrapped[a].H)):
                                twins[a] = b
                                break
            if sparsegpt and getattr(self.owner, "sparsegpt_factor_side_by_side", True):
                # the block's factorisations up front, one at a time (pruners/sparsegpt.py: why not side by side)
                with PhaseTimer.span("sparsegpt.factor (all Linears of a block, one at a time: csrc/cholesky.hip)"):
                    SparseGPT.factor_all([wrapped[n] for n in subset if n not in twins], percdamp=0.01)
            block_items = []
            for name in subset:
                assert wrapped[name].nsamples == sum(x.shape[0] for x in inps) * count_factor
                weight = subset[name].weight.data
                ratio = sparsity_ratio[f"{module_to_process}.{i}.{name}.weight"]
                if sparsegpt:           # sparsegpt_pruner.py:394 / :650
                    wrapped[name].fasterprune(ratio, prune_n=self.owner.prune_n,
                                              prune_m=self.owner.prune_m, percdamp=0.01,
                                              blocksize=128,
                                              same_hessian_as=wrapped.get(twins.get(name)))
                    wrapped[name].H = None
                elif getattr(self.owner, "prune_n", 0) != 0:   # structured n:m (:265-270 / :546-551)
                    self.kernels.wanda_prune_nm(weight, wrapped[name].scaler_row,
                                                self.owner.prune_n, self.owner.prune_m)
                elif mode == "rows":      # per output row, k smallest by stable order (:272-279)
                    block_items.append((weight, wrapped[name].scaler_row, "rows",
                                        int(weight.shape[1] * ratio), None))
                else:                   # whole matrix, metric <= sorted[k] (:555-558)
                    block_items.append((weight, wrapped[name].scaler_row, "matrix",
                                        int(weight.numel() * ratio), None))
            if block_items:
                # the block's selections are independent of each other (the reference's per-Linear
                # loop reads only that Linear's weight and statistic): one call, shared launches
                self.kernels.wanda_prune_block(block_items)
            if sparsegpt:
                for w_ in wrapped.values():
                    w_.free()
            with PhaseTimer.span("stage2.block_forward_after_prune"):
                if group_ok[0] and graph_pass_grouped(block):
                    pass
                elif graphed_plain:
                    group_ok[0] = False
                    graph_pass(block, None, keep=True)
                else:
                    for j in range(n_batches):
                        outs[j] = call(block, j)
            inps, outs = outs, inps
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        return model


def _uniform_calibration(inps, caches, n):
    """Same shapes / dtypes for every sample's block input and cached kwargs, equal non-tensor
    kwargs — what one captured graph can replay."""
    def sig(x, c):
        out = [tuple(x.shape), x.dtype]
        for k in sorted(c):
            v = c[k]
            if torch.is_tensor(v):
                out.append((k, tuple(v.shape), v.dtype, v.is_cuda))
            elif v is None or isinstance(v, (bool, int, float, str)):
                out.append((k, v))
            else:
                return None
        return tuple(out)
    first = sig(inps[0], caches[0])
    if first is None:
        return False
    for j in range(1, n):
        if not torch.is_tensor(inps[j]) or sig(inps[j], caches[j]) != first:
            return False
    return all(v.is_cuda for v in caches[0].values() if torch.is_tensor(v))


def _t5_block_mapping(names, granularity, depth=4):
    if granularity == "layer":
        return {k: k for k in names}
    if granularity == "block":
        return {k: ".".join(k.split(".")[:depth]) for k in names}
    raise NotImplementedError


class _StageOneMixin:
    def _hook_plan(self):
        """(block ModuleList paths, extra cacheable module paths) for a model without
        `stage_plan()`: the lists this pruner walks in stage 2, in execution order, plus the
        model's other direct children (Q-Former, projections, norms ...)."""
        lists = []
        for cand in (f"{getattr(self, 'vit_model_prefix', None)}.blocks",
                     f"{getattr(self, 'model_prefix', None)}.blocks",
                     f"{getattr(self, 't5_model_prefix', None)}.encoder.block",
                     f"{getattr(self, 't5_model_prefix', None)}.decoder.block",
                     f"{getattr(self, 'model_prefix', None)}.encoder.block",
                     f"{getattr(self, 'model_prefix', None)}.decoder.block"):
            try:
                mod = get_module_recursive(self.model, cand)
            except AttributeError:
                continue
            if isinstance(mod, nn.ModuleList) and cand not in lists:
                lists.append(cand)
        roots = {p.split(".")[0] for p in lists}
        extra = [n for n, _ in self.model.named_children() if n not in roots]
        return lists, extra

    def _eval_batch(self):
        """Evaluations of a layer per pass of the shared suffix.  What the caller set, or — 0, the
        default — sized from the calibration set: a layer has 2 x (its units on this rank)
        evaluations, and a pass should carry all of them up to ~256 samples (16 evaluations of
        batch-8 pairs were this build's tuning point: 128 samples; the launchers' batch size 1
        with 32 calibration samples leaves a 16-evaluation pass at 16 samples, where the suffix
        is launch-bound — round 6: ECoFLaP + SparseGPT's stage 1 48.6 s at 16, 33.0 s at 32,
        24.5 s at 64, 39.6 s at 128 (half the slots padding), the same table every time).  Powers
        of two in [16, 64]; the result does not depend on it (every pass is checked batch
        invariant, pruners/prefix_cache.py)."""
        eb = int(getattr(self, "eval_batch", 1))
        if eb:
            return eb
        try:
            first = self.data_loader[0]
            n = int(first["image"].shape[0]) if isinstance(first, dict) and "image" in first \
                else int(_default_batch_len(first))
        except Exception:
            return 16
        n = max(n, 1)
        units = -(-int(self.num_data_first_stage) // n) * max(int(self.num_noise), 1)
        import torch.distributed as dist
        world = dist.get_world_size(self.process_group) if dist.is_available() and dist.is_initialized() else 1
        evals = 2 * -(-units // world)
        upper = min(64, max(16, 256 // n))
        eb = 16
        while eb < evals and eb < upper:
            eb *= 2
        return eb

    def _layer_sparsity(self, loss_func, original_sparsity, mapping, per_model_group=()):
        if (getattr(self, "prefix_cache", True) and not hasattr(self.model, "stage_plan")
                and str(self.score_method).startswith("MEZO") and mapping):
            # un-staged model (the reference's own modules): exact suffix-only re-forward through
            # forward patches on the block lists (pruners/hooked_prefix.py)
            lists, extra = self._hook_plan()
            if lists:
                from .hooked_prefix import HookedPrefixLoss
                # (the lock-step path runs one Python thread per evaluation: it stays at 16)
                loss_func = HookedPrefixLoss(self.model, loss_func, lists, extra,
                                             eval_batch=min(self._eval_batch(), 16)
                                             if not int(getattr(self, "eval_batch", 1)) else self._eval_batch())
        if (getattr(self, "prefix_cache", True) and hasattr(self.model, "stage_plan")
                and str(self.score_method).startswith("MEZO") and mapping):
            # same losses, bit for bit, from the owning block onwards only
            from .prefix_cache import PrefixCachedLoss
            on_gpu = next(iter(self.model.parameters())).device.type == "cuda"
            loss_func = PrefixCachedLoss(
                self.model, kind="vision" if loss_func is loss_vision else "vision_language",
                use_graphs=bool(getattr(self, "use_graphs", True)) and on_gpu,
                n_lanes=int(getattr(self, "n_lanes", 2)),
                eval_batch=self._eval_batch())
        ls = LayerSparsity(
            self.model, self.data_loader, loss_func, self.num_data_first_stage, original_sparsity,
            self.max_sparsity_per_layer, self.score_method, self.num_noise, self.noise_eps,
            mapping, prune_per_model=self.prune_per_model, per_model_group=list(per_model_group),
            kernels=self.kernels, z_source=self.z_source, process_group=self.process_group,
            k1_form=getattr(self, "k1_form", "block"),
            grad_graphs=bool(getattr(self, "use_graphs", True)),
            checkpoint_path=getattr(self, "stage1_checkpoint", None))
        self.kernels = ls.kernels
        out = ls.return_sparsity()
        self.stage_stats["stage1"] = dict(ls.stats)
        self.last_loss_table = ls.loss_table      # [units, 2] fp32 (zeroth order; None otherwise)
        if getattr(ls, "resumed_layers", 0):
            self.stage_stats["stage1"]["resumed_layers"] = ls.resumed_layers
        if hasattr(loss_func, "stats"):
            self.stage_stats["stage1"]["suffix_forward"] = {
                k: v for k, v in loss_func.stats.items() if k != "stages_not_batch_invariant"}
            names = sorted(loss_func.stats.get("stages_not_batch_invariant", []))
            self.stage_stats["stage1"]["stages_not_batch_invariant"] = len(names)
            self.stage_stats["stage1"]["stages_not_batch_invariant_names"] = names[:4] + names[-2:] \
                if len(names) > 6 else names
        self.layer_sparsity_engine = ls
        return out

    def _load_sparsity_yaml(self):
        import yaml
        with open(self.sparsity_dict, "r") as f:
            return yaml.load(f, Loader=yaml.FullLoader)


@registry.register_pruner("t5_wanda_pruner")
class T5LayerWandaPruner(_StageOneMixin, LayerWiseBasePruner):
    pruner_name = "t5_wanda_pruner"

    def __init__(self, model, data_loader, model_prefix="t5_model", **kwargs):
        super().__init__(model=model, data_loader=data_loader, model_prefix=model_prefix, **kwargs)
        self.loss_func = loss_language

    def forward_to_cache(self, model, batch):
        return model(batch)

    @print_time
    def _prune(self, model, dataloader, device, model_prefix, module_to_process="encoder.block",
               n_samples=64, sparsity_ratio=0.5):
        cfg = getattr(model, model_prefix).config
        use_cache, cfg.use_cache = cfg.use_cache, False
        try:
            _BlockwiseWanda(self).run(
                model, dataloader, module_to_process, n_samples, sparsity_ratio,
                forward_fn=lambda m, b: self.forward_to_cache(m, b), cache_keys=T5_BLOCK_KWARGS,
                autocast=lambda: model.maybe_autocast(dtype=torch.bfloat16), take_first=True,
                mode="rows")
        finally:
            cfg.use_cache = use_cache
        return model

    def get_sparsity(self, original_sparsity, sparsity_ratio_granularity=None):
        if self.sparsity_dict is not None:
            return self._load_sparsity_yaml()
        if sparsity_ratio_granularity is None:
            mapping = {}
        else:
            names = [k for k, v in self.model.named_parameters()
                     if len(v.shape) == 2 and ".block" in k
                     and "relative_attention_bias.weight" not in k
                     and k.startswith(self.model_prefix)]
            mapping = _t5_block_mapping(names, sparsity_ratio_granularity, depth=4)
        return self._layer_sparsity(loss_language, original_sparsity, mapping)

    @print_time
    def prune(self, importance_scores=None, keep_indices_or_masks=None):
        print("In: ", self.pruner_name)
        dtype_record, requires_grad_record, device = self.model_setup_and_record_attributes(self.model)
        if self.prune_spec is None:
            return self.model, None
        _, keep_ratio, _, _ = self.convert_spec_to_list(self.prune_spec)
        sparsity_dict = self.get_sparsity(1 - keep_ratio,
                                          sparsity_ratio_granularity=self.sparsity_ratio_granularity)
        for part in ("encoder", "decoder"):
            self.model = self._prune(self.model, self.data_loader, device,
                                     model_prefix=self.model_prefix,
                                     module_to_process=f"{self.model_prefix}.{part}.block",
                                     n_samples=self.num_samples, sparsity_ratio=sparsity_dict)
        self.model_reset(self.model, dtype_record, requires_grad_record, device)
        return self.model, sparsity_dict


@registry.register_pruner("vit_wanda_pruner")
class VITLayerWandaPruner(_StageOneMixin, LayerWiseBasePruner):
    pruner_name = "vit_wanda_pruner"

    def __init__(self, model, data_loader, model_prefix="visual", **kwargs):
        super().__init__(model=model, data_loader=data_loader, model_prefix=model_prefix, **kwargs)
        self.loss_func = loss_vision

    def forward_to_cache(self, model, batch):
        return model.encode_image(batch["image"])

    @print_time
    def _prune(self, model, dataloader, device, model_prefix, module_to_process="encoder.block",
               n_samples=64, sparsity_ratio=0.5):
        return _BlockwiseWanda(self).run(
            model, dataloader, module_to_process, n_samples, sparsity_ratio,
            forward_fn=lambda m, b: self.forward_to_cache(m, b), cache_keys=["rel_pos_bias"],
            autocast=lambda: model.maybe_autocast(), take_first=False, mode="matrix")

    def get_sparsity(self, original_sparsity, sparsity_ratio_granularity=None):
        if self.sparsity_dict is not None:
            sparsity_dict = self._load_sparsity_yaml()
            # tables written by the multi-modal pruner use the BLIP-2 prefix (:576-583)
            sparsity_dict = {k.replace("visual_encoder.", "visual."): v
                             for k, v in sparsity_dict.items()}
            if "visual.blocks.39.attn.qkv.weight" not in sparsity_dict:
                for leaf in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
                    sparsity_dict[f"visual.blocks.39.{leaf}.weight"] = 0
            return sparsity_dict
        if sparsity_ratio_granularity is None:
            mapping = {}
        else:
            names = [k for k, v in self.model.named_parameters()
                     if len(v.shape) == 2 and ".blocks" in k and k.startswith(self.model_prefix)]
            mapping = _t5_block_mapping(names, sparsity_ratio_granularity, depth=3)
        return self._layer_sparsity(loss_vision, original_sparsity, mapping)

    @print_time
    def prune(self, importance_scores=None, keep_indices_or_masks=None):
        print("In: ", self.pruner_name)
        dtype_record, requires_grad_record, device = self.model_setup_and_record_attributes(self.model)
        if self.prune_spec is None:
            return self.model, None
        _, keep_ratio, _, _ = self.convert_spec_to_list(self.prune_spec)
        sparsity_dict = self.get_sparsity(1 - keep_ratio,
                                          sparsity_ratio_granularity=self.sparsity_ratio_granularity)
        self.model = self._prune(self.model, self.data_loader, device,
                                 model_prefix=self.model_prefix,
                                 module_to_process=f"{self.model_prefix}.blocks",
                                 n_samples=self.num_samples, sparsity_ratio=sparsity_dict)
        self.model_reset(self.model, dtype_record, requires_grad_record, device)
        return self.model, sparsity_dict


@registry.register_pruner("blipt5_wanda_pruner")
class BLIPT5LayerWandaPruner(_StageOneMixin, LayerWiseBasePruner):
    pruner_name = "blipt5_wanda_pruner"

    def __init__(self, model, data_loader, t5_prune_spec=None, vit_prune_spec=None,
                 t5_pruning_method=None, vit_pruning_method=None, t5_model_prefix="t5_model",
                 vit_model_prefix="visual_encoder", **kwargs):
        kwargs.pop("prune_spec", None)
        kwargs.pop("model_prefix", None)
        super().__init__(model=model, data_loader=data_loader, prune_spec=None,
                         model_prefix="tmp", **kwargs)
        self.t5_prune_spec = t5_prune_spec
        self.vit_prune_spec = vit_prune_spec
        assert t5_pruning_method is not None
        assert vit_pruning_method is not None
        self.t5_model_prefix = t5_model_prefix
        self.vit_model_prefix = vit_model_prefix

    def forward_to_cache(self, model, batch):
        return model(batch)

    def get_sparsity(self, original_sparsity, sparsity_ratio_granularity=None):
        if self.sparsity_dict is not None:
            return self._load_sparsity_yaml()
        t5p, vitp = self.t5_model_prefix, self.vit_model_prefix
        if sparsity_ratio_granularity is None:
            mapping = {}
        else:
            names = [k for k, v in self.model.named_parameters()
                     if len(v.shape) == 2 and ".block" in k
                     and "relative_attention_bias.weight" not in k
                     and (k.startswith(t5p) or k.startswith(vitp))]

            def group_of(name):
                is_t5 = name.startswith(t5p)
                if sparsity_ratio_granularity == "model":
                    return t5p if is_t5 else vitp
                if sparsity_ratio_granularity == "layer":
                    return name
                if sparsity_ratio_granularity == "block":
                    return ".".join(name.split(".")[:4 if is_t5 else 3])
                raise NotImplementedError

            mapping = {k: group_of(k) for k in names}
        return self._layer_sparsity(loss_vision_language, original_sparsity, mapping,
                                    per_model_group=[t5p, vitp])

    def _vit_prune(self, model, dataloader, device, model_prefix, module_to_process, n_samples,
                   sparsity_ratio):
        return _BlockwiseWanda(self).run(
            model, dataloader, module_to_process, n_samples, sparsity_ratio,
            forward_fn=lambda m, b: self.forward_to_cache(m, b), cache_keys=["rel_pos_bias"],
            autocast=lambda: model.maybe_autocast(), take_first=False, mode="matrix")

    def _t5_prune(self, model, dataloader, device, model_prefix, module_to_process, n_samples,
                  sparsity_ratio):
        cfg = getattr(model, model_prefix).config
        use_cache, cfg.use_cache = cfg.use_cache, False
        try:
            _BlockwiseWanda(self).run(
                model, dataloader, module_to_process, n_samples, sparsity_ratio,
                forward_fn=lambda m, b: self.forward_to_cache(m, b), cache_keys=T5_BLOCK_KWARGS,
                autocast=lambda: model.maybe_autocast(dtype=torch.bfloat16), take_first=True,
                mode="rows")
        finally:
            cfg.use_cache = use_cache
        return model

    @print_time
    def prune(self, importance_scores=None, keep_indices_or_masks=None):
        print("In: ", self.pruner_name)
        dtype_record, requires_grad_record, device = self.model_setup_and_record_attributes(self.model)

        global_sparsity_dict = None
        if self.sparsity_ratio_granularity is not None:
            _, vit_keep_ratio, _, _ = self.convert_spec_to_list(self.vit_prune_spec)
            _, t5_keep_ratio, _, _ = self.convert_spec_to_list(self.t5_prune_spec)
            assert vit_keep_ratio == t5_keep_ratio
            global_sparsity_dict = self.get_sparsity(
                1 - vit_keep_ratio, sparsity_ratio_granularity=self.sparsity_ratio_granularity)

        def table_for(spec):
            if global_sparsity_dict is not None:
                return global_sparsity_dict
            _, keep_ratio, _, _ = self.convert_spec_to_list(spec)
            return self.get_sparsity(1 - keep_ratio, sparsity_ratio_granularity=None)

        if self.vit_prune_spec is not None:
            self.model = self._vit_prune(
                self.model, self.data_loader, device, model_prefix=self.vit_model_prefix,
                module_to_process=f"{self.vit_model_prefix}.blocks", n_samples=self.num_samples,
                sparsity_ratio=table_for(self.vit_prune_spec))
        if self.t5_prune_spec is not None:
            table = table_for(self.t5_prune_spec)
            for part in ("encoder", "decoder"):
                self.model = self._t5_prune(
                    self.model, self.data_loader, device, model_prefix=self.t5_model_prefix,
                    module_to_process=f"{self.t5_model_prefix}.{part}.block",
                    n_samples=self.num_samples, sparsity_ratio=table)
        self.model_reset(self.model, dtype_record, requires_grad_record, device)
        return self.model, global_sparsity_dict
