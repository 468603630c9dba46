"""Exact suffix-only re-forward for a model that does NOT expose `stage_plan()` — the
reference's own LAVIS / UPop modules swapped in per INTEGRATION.md §A.

`PrefixCachedLoss` (prefix_cache.py) needs the model's forward as a composition of stage
functions.  This adapter gets the same effect from the outside, with nothing but the module
tree: the block `nn.ModuleList`s the pruner walks anyway (`visual_encoder.blocks`,
`t5_model.encoder.block`, `t5_model.decoder.block`, ...) and, optionally, other sub-modules that
run before them (`Qformer`, `ln_vision`, `t5_proj`, ...).

  * A forward of the model is a fixed sequence of calls of those "cacheable" modules (eval mode,
    same batch).  The first forward of a calibration batch records that sequence and each call's
    output.
  * Layers are scored in parameter order and never change again, so while a matrix of module M
    is perturbed every cacheable call that happens BEFORE M's first call has exactly the output it
    had last time.  For those calls `forward` is replaced by a function that hands back the
    recorded output (a clone: downstream code may write in place); M and everything after it —
    and all the model's glue code in between — runs for real.
  * When the scored matrix moves on to a later module, the calls between the old and the new
    owner are simply run (and recorded) once more by the first evaluation that needs them, with
    the weights the finished layers were left with.

The loss is therefore the loss of a full forward, bit for bit (same kernels on the same bits
from the owning module on; everything before it is the very tensor a full forward would
recompute), at the cost of the owning module's suffix plus the un-cached glue.  No HIP graphs
and no lanes: the modules are opaque, their launches stay eager.
"""
import torch


def _map(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        out = [_map(v, fn) for v in obj]
        return type(obj)(out) if not hasattr(obj, "_fields") else type(obj)(*out)
    return obj


def _resolve(model, path):
    mod = model
    for part in [p for p in path.split(".") if p]:
        mod = getattr(mod, part)
    return mod


class _SequenceChanged(Exception):
    pass


class HookedPrefixLoss:
    """Drop-in `loss_func(model, samples, cuda_enabled) -> (loss, batch_len)` around another
    loss closure; `LayerSparsity` announces the scored matrix through `begin_layer(name)`."""

    def __init__(self, model, loss_func, block_lists, extra_modules=(), max_batches=256):
        self.model = model
        # batches whose recorded activations are kept (least recently used goes first).
        # `LayerSparsity` takes the calibration prefix ONCE and re-uses those batch objects for
        # every layer, so the working set is the number of calibration batches (128 at batch
        # size 1 for the reference's defaults); a loader that yields FRESH objects per pass gets
        # no reuse out of an identity-keyed cache and must not grow it without bound either
        self.max_batches = int(max_batches)
        self._held = {}                       # id(samples) -> samples, in LRU order
        self.loss_func = loss_func
        self.paths = {}                       # module -> dotted path
        for lp in block_lists:
            blocks = _resolve(model, lp)
            for i, blk in enumerate(blocks):
                self.paths[blk] = f"{lp}.{i}"
        for ep in extra_modules:
            self.paths[_resolve(model, ep)] = ep
        self.by_prefix = sorted(((p + ".", m) for m, p in self.paths.items()),
                                key=lambda t: -len(t[0]))
        self.owner = None                     # module owning the matrix being scored
        self.sequence = None                  # [module] in call order (one forward)
        self.cache = {}                       # id(samples) -> [outputs per event]; valid prefix
        self.valid = {}                       # id(samples) -> number of leading events recorded
        self.disabled = False                 # the model does not call its modules in a fixed order
        self.stats = {"events_total": 0, "events_served": 0, "forwards": 0}

    # ---- hooks of LayerSparsity ------------------------------------------------------------
    def begin_layer(self, name):
        owner = None
        for prefix, mod in self.by_prefix:
            if name.startswith(prefix):
                owner = mod
                break
        if owner is not self.owner:
            # outputs of the old owner and of everything after it were never recorded while it
            # was being perturbed (see `limit` below), so nothing has to be thrown away
            self.owner = owner

    def reset(self):
        self.cache.clear()
        self.valid.clear()
        self._held.clear()
        self.owner = None

    # ---- the loss closure ----------------------------------------------------------------------
    def _limit(self):
        """Events [0, limit) happen before the owner's first call: they may be served / recorded."""
        if self.sequence is None:
            return 0
        if self.owner is None:
            return 0                              # unknown owner: run everything, cache nothing
        for i, mod in enumerate(self.sequence):
            if mod is self.owner:
                return i
        return 0

    def __call__(self, model, samples, cuda_enabled):
        assert model is self.model
        if self.disabled:
            return self.loss_func(model, samples, cuda_enabled)
        try:
            return self._cached_call(model, samples, cuda_enabled)
        except _SequenceChanged:
            # a forward that calls its cacheable modules in another order / number than the
            # recorded one (data-dependent control flow): the record cannot be trusted — drop it
            # for good and evaluate this and every later loss with plain full forwards
            import warnings
            warnings.warn("HookedPrefixLoss: the model's module call sequence is not fixed; "
                          "falling back to full forwards")
            self.disabled = True
            self.cache.clear()
            self.valid.clear()
            return self.loss_func(model, samples, cuda_enabled)

    def _cached_call(self, model, samples, cuda_enabled):
        key = id(samples)
        # the cache is keyed by the batch OBJECT: keep it alive (an id can be recycled once its
        # object is gone — a loader yielding fresh batches would be served another batch's
        # activations) and check identity
        held = self._held
        if held.get(key) is not samples:
            self.cache.pop(key, None)
            self.valid.pop(key, None)
        held.pop(key, None)
        held[key] = samples                       # most recently used last
        while len(held) > self.max_batches:
            old = next(iter(held))
            del held[old]
            self.cache.pop(old, None)
            self.valid.pop(old, None)
            self.stats["evicted"] = self.stats.get("evicted", 0) + 1
        record_sequence = self.sequence is None
        seen = []
        limit = self._limit()
        cached = self.cache.setdefault(key, [])
        n_valid = min(self.valid.get(key, 0), limit)
        counter = [0]
        originals = {}

        def make(mod, real):
            def patched(*args, **kwargs):
                i = counter[0]
                counter[0] += 1
                if record_sequence:
                    seen.append(mod)
                elif i >= len(self.sequence) or self.sequence[i] is not mod:
                    raise _SequenceChanged()
                if i < n_valid:
                    self.stats["events_served"] += 1
                    return _map(cached[i], lambda t: t.clone())
                out = real(*args, **kwargs)
                if i < limit or record_sequence:
                    # (first forward of all: record everything once to learn the sequence; only
                    # the events before the owner count as valid)
                    keep = _map(out, lambda t: t.detach().clone())
                    if i < len(cached):
                        cached[i] = keep
                    elif i == len(cached):
                        cached.append(keep)
                    else:
                        raise _SequenceChanged()
                return out
            return patched

        had_own = {}
        for mod in self.paths:
            had_own[mod] = "forward" in mod.__dict__
            originals[mod] = mod.forward
            mod.forward = make(mod, originals[mod])
        try:
            out = self.loss_func(model, samples, cuda_enabled)
        finally:
            for mod, fwd in originals.items():
                if had_own[mod]:
                    mod.forward = fwd
                else:
                    del mod.forward       # back to the class's forward: nothing left on the instance

        if record_sequence:
            self.sequence = seen
            limit = self._limit()
        elif counter[0] != len(self.sequence):
            raise _SequenceChanged()
        self.valid[key] = min(max(self.valid.get(key, 0), limit), len(cached))
        del cached[self.valid[key]:]              # nothing recorded at / after the owner survives
        self.stats["events_total"] += counter[0]
        self.stats["forwards"] += 1
        return out
