"""Exact suffix-only re-forward for a model that does NOT expose `stage_plan()` — the
reference's own LAVIS / UPop modules swapped in per INTEGRATION.md §A — and, since round 5, the
batched evaluation of a layer's units on such a model.

`PrefixCachedLoss` (prefix_cache.py) needs the model's forward as a composition of stage
functions.  This adapter gets the same effect from the outside, with nothing but the module
tree: the block `nn.ModuleList`s the pruner walks anyway (`visual_encoder.blocks`,
`t5_model.encoder.block`, `t5_model.decoder.block`, ...) and, optionally, other sub-modules that
run before them (`Qformer`, `ln_vision`, `t5_proj`, ...).

  * A forward of the model is a fixed sequence of calls ("events") of those cacheable modules
    (eval mode, same batch).  The first forward of a calibration batch records that sequence,
    each call's output, the autocast state it ran under, and its WIRING: whether every tensor it
    was handed is, by object identity, an output or an argument of the previous event (the glue
    between two block calls of a `for blk in self.blocks` loop is pure plumbing).
  * Layers are scored in parameter order and never change again, so while a matrix of module M
    is perturbed every event BEFORE M's first call has exactly the output it had last time:
    those calls are served from the record; M and everything after it runs for real.
  * `multi()` (the protocol `LayerSparsity` uses for `PrefixCachedLoss` too) evaluates the k =
    `eval_batch` evaluations of a chunk in LOCK STEP: each evaluation's `loss_func(model, batch)`
    runs in a thread of its own, one thread at a time (a baton, no concurrency: threads are only
    how a Python forward is suspended in the middle of the model's own glue code).  A thread that
    reaches a not-yet-computed event parks there with the arguments the model handed it; when all
    k are parked, the coordinator runs that event ONCE on the k argument sets concatenated along
    dim 0 — and, following the wiring, every event after it up to the next piece of real glue —
    and the threads resume with their slots.  The owning block runs at batch k*B with the
    perturbed Linear alone applied per slot (theta straight from K1's scratch).  So the blocks
    behind the owner run once per chunk instead of k times, and only the model's glue (embeddings,
    masks, the loss head) still runs per evaluation.
  * Exactness is measured, not assumed, exactly as in prefix_cache.py: an event is shared only if
    its batched result equals, bit for bit, what one evaluation gets alone (probed on first use
    per batch-shape family and width); every matrix's per-slot owner is checked against the
    per-evaluation call on first use; one evaluation per entry block, its slot rotating, is
    re-evaluated sequentially and its loss compared bit for bit.  Anything that fails falls back
    to the per-evaluation path (always correct) for that event, or for the run.

The loss is therefore the loss of a full forward, bit for bit.  No HIP graphs: the modules are
opaque, their launches stay eager.
"""
import contextlib
import os
import threading
import time

import torch

_TLS = threading.local()


def _map(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        out = [_map(v, fn) for v in obj]
        return type(obj)(out) if not hasattr(obj, "_fields") else type(obj)(*out)
    return obj


def _flatten(obj):
    """-> (leaves, spec): tensors and everything that is not a dict / list / tuple are leaves."""
    leaves = []

    def walk(o):
        if torch.is_tensor(o):
            leaves.append(o)
            return ("t", len(leaves) - 1)
        if isinstance(o, dict):
            return ("d", type(o), tuple((k, walk(v)) for k, v in o.items()))
        if isinstance(o, (list, tuple)):
            return ("s", type(o), tuple(walk(v) for v in o))
        leaves.append(o)
        return ("c", len(leaves) - 1)
    return leaves, walk(obj)


def _unflatten(spec, leaves):
    kind = spec[0]
    if kind in ("t", "c"):
        return leaves[spec[1]]
    if kind == "d":
        items = [(k, _unflatten(s, leaves)) for k, s in spec[2]]
        if spec[1] is dict:
            return dict(items)
        try:
            return spec[1](items)
        except Exception:
            try:
                return spec[1](**dict(items))
            except Exception:
                return dict(items)
    seq = [_unflatten(s, leaves) for s in spec[2]]
    T = spec[1]
    return T(*seq) if hasattr(T, "_fields") else T(seq)


def _resolve(model, path):
    mod = model
    for part in [p for p in path.split(".") if p]:
        mod = getattr(mod, part)
    return mod


def _autocast_state():
    return (torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda"),
            torch.is_autocast_enabled("cpu"), torch.get_autocast_dtype("cpu"))


@contextlib.contextmanager
def _autocast_as(state):
    with contextlib.ExitStack() as st:
        if torch.cuda.is_available():
            st.enter_context(torch.autocast("cuda", dtype=state[1], enabled=state[0]))
        st.enter_context(torch.autocast("cpu", dtype=state[3], enabled=state[2]))
        yield


class _SequenceChanged(Exception):
    pass


class _NotBatchLeading(Exception):
    pass


class _Abort(BaseException):
    """Unwinds a worker's forward once the coordinator has given the chunk up."""


class _Ctx:
    """One forward in flight: which batch, what may be served, where it is in the sequence."""
    __slots__ = ("owner", "key", "n_valid", "limit", "cached", "counter", "ready", "expect", "pending",
                 "lock", "seen", "record", "abort", "wake", "result", "error", "done", "slot")

    def __init__(self, owner, key, n_valid, limit, cached, lock):
        self.owner, self.key, self.n_valid, self.limit, self.cached = owner, key, n_valid, limit, cached
        self.counter = 0
        self.ready, self.expect, self.pending = {}, {}, None
        self.lock = lock              # lock-step worker (parks at heads) or a plain sequential forward
        self.seen, self.record = [], None
        self.abort, self.wake, self.result, self.error, self.done = False, None, None, None, False
        self.slot = 0


class _Worker(threading.Thread):
    """One lock-step evaluation at a time; lives as long as the closure that started it."""

    def __init__(self):
        super().__init__(daemon=True)
        self.wake = threading.Event()
        self.job = None
        self.stop = False

    def run(self):
        while True:
            self.wake.wait()
            if self.stop:
                return
            job, self.job = self.job, None
            if job is None:                       # nothing to do: wait for the next start
                self.wake.clear()
                continue
            job()                                 # (clears `wake` itself before it hands over)


class HookedPrefixLoss:
    """Drop-in `loss_func(model, samples, cuda_enabled) -> (loss, batch_len)` around another
    loss closure; `LayerSparsity` announces the scored matrix through `begin_layer(name)` and,
    with eval_batch > 1, hands whole chunks of evaluations to `multi()`."""

    requires_static_weights = False       # no graphs: theta is re-pointed / applied per slot

    def __init__(self, model, loss_func, block_lists, extra_modules=(), max_batches=256, eval_batch=1,
                 verify_batched="entries", use_graphs=True, defer_guard=False):
        self.model = model
        # batches whose recorded activations are kept (least recently used goes first).
        # `LayerSparsity` takes the calibration prefix ONCE and re-uses those batch objects for
        # every layer, so the working set is the number of calibration batches (it says so through
        # `set_working_set`); a loader that yields FRESH objects per pass gets no reuse out of an
        # identity-keyed cache and must not grow it without bound either
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
        self.wired = {}                       # event -> (spec, refs) when its arguments are plumbing
        self.autocast_at = {}                 # event -> autocast state at its call site
        self.cache = {}                       # id(samples) -> [outputs per event]; valid prefix
        self.valid = {}                       # id(samples) -> number of leading events recorded
        self.blen = {}                        # id(samples) -> batch length the loss closure reported
        self.disabled = False                 # the model does not call its modules in a fixed order
        assert eval_batch >= 1 and (eval_batch == 1 or eval_batch % 2 == 0)
        self.eval_batch = int(eval_batch)
        assert verify_batched in ("entries", "all")
        self.verify_batched = verify_batched
        self.invariant = {}                   # (family, width, event) -> shared result == alone, bit for bit
        self._owner_ok = {}                   # (family, matrix) -> per-slot owner == per-evaluation owner
        self._owner_block_ok = {}             # (family, owner event, width) -> checked on every slot
        # inside a wired run of blocks the parked forwards get the un-sliced batched output back
        # (they only hand it on to the next, served, block); False: every event sliced per evaluation
        self.lazy_slices = os.environ.get("ECOFLAP_LOCKSTEP_LAZY", "1") != "0"
        self._verified = set()                # (family, owner event) whose lock-step losses were checked
        self._pair_name = self._pair_home = self._pair_param = None
        # the shared (batched) calls of the cacheable modules behind the owner are replayed from HIP
        # graphs when a module lets itself be captured (static shapes, no host sync); one that does
        # not stays eager.  The model's own glue and the owning block are never captured.
        self.use_graphs = bool(use_graphs) and os.environ.get("ECOFLAP_LOCKSTEP_GRAPHS", "1") != "0"
        self._graphs = {}                     # (family, width, event, pad) -> captured call or False
        self._graph_pool = None
        self._workers = []                    # lock-step threads, kept from chunk to chunk
        self._params = None                   # name -> Parameter (built once: 1000+ entries)
        self._keep_patched = False            # inside a LayerSparsity pass
        self._patch_state = None
        self._lens = {}                       # id(samples) -> batch length, for `multi`'s result
        self._value_shared = set()            # (family, event, leaf) tensors equal across evaluations
        self._assumed = None                  # device flag: such a tensor differed after all
        # defer_guard (or ECOFLAP_LOCKSTEP_DEFER_GUARD=1): the per-entry-block guard compares on the
        # device and is read back without stopping the host (round 6; `_guard_deferred`).  OFF by
        # default: measured on the bench's matrices it takes 4-5 ms of blocked time off a 100 ms
        # step that is bound by the host's own 72-80 ms (9.8-10.2 against 9.5-10.1 layers/s, config
        # 3 inside the box-to-box spread; profiles/NOTES_r06.md section 6), and a difference found
        # a chunk late can only be an error, where the synchronous form falls back to
        # per-evaluation forwards and carries on
        self.defer_guard = bool(defer_guard) or os.environ.get("ECOFLAP_LOCKSTEP_DEFER_GUARD", "0") == "1"
        self._probes = []                     # (event on the read-back stream, pinned host copy of the flag, what)
        self._probe_stream = None
        self.stats = {"events_total": 0, "events_served": 0, "forwards": 0}

    # ---- hooks of LayerSparsity ------------------------------------------------------------
    def _owner_of(self, name):
        for prefix, mod in self.by_prefix:
            if name.startswith(prefix):
                return mod
        return None

    def begin_layer(self, name):
        owner = self._owner_of(name)
        if owner is not self.owner:
            # outputs of the old owner and of everything after it were never recorded while it
            # was being perturbed (see `limit` below), so nothing has to be thrown away
            self.owner = owner

    def stage_of(self, name):
        """Layers owned by one block share a K1 launch (`LayerSparsity`, k1_form="block")."""
        owner = self._owner_of(name)
        return self.paths[owner] if owner is not None else ("?", name)

    def set_working_set(self, n_batches):
        """`LayerSparsity` knows how many batch objects it cycles through: an LRU smaller than
        that would evict every batch right before its next use."""
        self.max_batches = max(self.max_batches, int(n_batches))

    def close(self):
        """End the lock-step worker threads (they are daemons: not calling this leaks nothing
        past the interpreter's exit)."""
        self._keep_patched = False
        self._uninstall()
        for w in self._workers:
            w.stop = True
            w.wake.set()
        self._workers = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.cache.clear()
        self.valid.clear()
        self._held.clear()
        self.owner = None

    # ---- the pair protocol (LayerSparsity's batched path) ---------------------------------------
    def supports_pairs(self):
        return self.eval_batch > 1 and not self.disabled

    def pairs_in_flight(self):
        return max(1, self.eval_batch // 2)

    def begin_layer_weights(self, name, home):
        self._pair_name, self._pair_home = name, home
        self._keep_patched = True
        if self._params is None:
            self._params = dict(self.model.named_parameters())
        self._pair_param = self._params[name]

    def end_layer_weights(self, final):
        self._pair_name = self._pair_home = self._pair_param = None

    def join(self):
        pass

    def abort_run(self):
        """`LayerSparsity` calls this when a pass dies: the instances get their forwards back."""
        self._keep_patched = False
        self._uninstall()
        self._probes = []

    def finish_run(self):
        """Called by `LayerSparsity` before it reads the loss table: the instances get their
        forwards back and the assumptions made without a host sync must have held."""
        self._keep_patched = False
        self._uninstall()
        self._probes = []             # (the blocking read below covers whatever they would have shown)
        self.check_assumed()

    def _guard_deferred(self, got, want, what):
        """The guard's comparison without stopping the host: one compare launch ORs into the flag
        `check_assumed` reads (before every stage-1 checkpoint and at the end of the run), and a
        copy of the flag travels to pinned host memory on a side stream behind it; `_poll_assumed`
        — at the start of every later chunk — looks at the copies that have ARRIVED (an event
        query, no wait) and raises on the first non-zero one, i.e. a chunk or two after the
        difference happened.  Why: the host-side `torch.equal` drained the launch queue once per
        entry block (and `check_assumed` behind it again), after which the device idled until the
        host had refilled it — a quarter of the un-staged path's step was spent in those syncs
        (profiles/NOTES_r06.md section 6).  What it costs: a difference is an ERROR here ("rerun
        with eval_batch=1"), as for every other check queued without a sync; the graceful
        fall-back of the synchronous form (`defer_guard=False`) cannot be offered a chunk late."""
        from .prefix_cache import _differ_flag
        self._assumed = _differ_flag([t.reshape(1) for t in got], [t.reshape(1) for t in want], self._assumed)
        self._assumed_what = what
        done_main = torch.cuda.Event()
        done_main.record()
        if self._probe_stream is None:
            self._probe_stream = torch.cuda.Stream()
        host = torch.zeros(1, dtype=torch.int32).pin_memory()
        with torch.cuda.stream(self._probe_stream):
            self._probe_stream.wait_event(done_main)
            host.copy_(self._assumed, non_blocking=True)
            arrived = torch.cuda.Event()
            arrived.record(self._probe_stream)
        self._probes.append((arrived, host, what))
        self.stats["lockstep_checks_deferred"] = self.stats.get("lockstep_checks_deferred", 0) + 1

    def _poll_assumed(self):
        while self._probes and self._probes[0][0].query():
            _, host, what = self._probes.pop(0)
            if int(host[0]) != 0:
                self._probes = []
                raise RuntimeError("HookedPrefixLoss: a check that was queued without a host sync failed — last "
                                   f"queued when it was read: {what}; rerun with eval_batch=1")

    def check_assumed(self):
        """Read the checks that were queued without a host sync (one sync); raises if one failed.
        No other effect: the forward patches stay as they are — `LayerSparsity` calls this before
        every stage-1 checkpoint it writes, the lock-step guard after every entry block."""
        if self._assumed is not None and bool(self._assumed.item()):
            raise RuntimeError("HookedPrefixLoss: a check that was queued without a host sync failed (a "
                               "tensor assumed equal across the evaluations of a chunk differed, or the "
                               f"per-slot form of a matrix — last queued: {getattr(self, '_assumed_what', None)} "
                               "— did not equal its per-evaluation call); rerun with eval_batch=1")

    # ---- bookkeeping ---------------------------------------------------------------------------
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

    def _touch(self, key, samples):
        # the cache is keyed by the batch OBJECT: keep it alive (an id can be recycled once its
        # object is gone — a loader yielding fresh batches would be served another batch's
        # activations) and check identity
        held = self._held
        if held.get(key) is not samples:
            self.cache.pop(key, None)
            self.valid.pop(key, None)
            self.blen.pop(key, None)
        held.pop(key, None)
        held[key] = samples                       # most recently used last
        while len(held) > self.max_batches:
            old = next(iter(held))
            del held[old]
            self.cache.pop(old, None)
            self.valid.pop(old, None)
            self.blen.pop(old, None)
            self.stats["evicted"] = self.stats.get("evicted", 0) + 1
            if self.stats["evicted"] == 1:
                import warnings
                warnings.warn(f"HookedPrefixLoss: more than {self.max_batches} calibration batches in "
                              "flight; the oldest records are dropped (their prefixes re-run)")

    def _install(self):
        """Instance-level forwards on the cacheable modules (written straight into the instance
        dict: nn.Module.__setattr__ costs 4 us a piece, 100 modules, every chunk)."""
        if self._patch_state is not None:
            return
        originals, had_own = {}, {}
        for mod in self.paths:
            had_own[mod] = "forward" in mod.__dict__
            originals[mod] = mod.forward
            mod.__dict__["forward"] = self._make_patch(mod, originals[mod])
        self._patch_state = (originals, had_own)
        self._real = originals

    def _uninstall(self):
        """The instances are left as they were found (deepcopy / pickle see plain modules)."""
        if self._patch_state is None:
            return
        originals, had_own = self._patch_state
        for mod, fwd in originals.items():
            if had_own[mod]:
                mod.__dict__["forward"] = fwd
            else:
                mod.__dict__.pop("forward", None)   # back to the class's forward
        self._patch_state = None
        self._real = None

    @contextlib.contextmanager
    def _patched(self):
        """Patches for the duration of a call; inside a `LayerSparsity` pass (between
        `begin_layer_weights` and `finish_run`) they stay on from chunk to chunk."""
        fresh = self._patch_state is None
        self._install()
        try:
            yield
        finally:
            if fresh and not self._keep_patched:
                self._uninstall()

    def _make_patch(self, mod, real):
        def patched(*args, **kwargs):
            ctx = getattr(_TLS, "ctx", None)
            if ctx is None or ctx.owner is not self:
                return real(*args, **kwargs)      # (the coordinator running an event itself)
            return self._on_event(ctx, mod, real, args, kwargs)
        return patched

    # ---- one event of one forward ----------------------------------------------------------------
    def _on_event(self, ctx, mod, real, args, kwargs):
        i = ctx.counter
        ctx.counter = i + 1
        seq = self.sequence
        if seq is None:
            ctx.seen.append(mod)
        elif i >= len(seq) or seq[i] is not mod:
            raise _SequenceChanged()
        if i < ctx.n_valid:
            self.stats["events_served"] += 1
            out = ctx.cached[i]
            # handed out as it is when the next event is served too and takes it by plumbing; a
            # consumer that computes gets a copy (it may write in place)
            if i + 1 < ctx.n_valid and (i + 1) in self.wired:
                return out
            return _map(out, lambda t: t.clone())
        if ctx.lock:
            if i not in ctx.ready:
                # head of a segment nobody has computed yet: park with what the model handed over
                ctx.pending = (i, mod, args, kwargs, _autocast_state())
                self._park(ctx)
            else:
                exp = ctx.expect.get(i)
                if exp is not None and args and torch.is_tensor(args[0]) and args[0] is not exp:
                    raise _SequenceChanged()      # the glue is not the plumbing it was recorded as
            return ctx.ready.pop(i)
        # sequential forward: run for real, record what lies before the owner
        recording = seq is None
        if recording:
            leaves, spec = _flatten((args, kwargs))
            prev = ctx.record
            if prev is not None:
                self._learn_wiring(i, leaves, spec, prev)
            self.autocast_at[i] = _autocast_state()
        out = real(*args, **kwargs)
        if recording:
            ctx.record = (leaves, _flatten(out)[0])
        if i < ctx.limit or recording:
            # (first forward of all: record everything once to learn the sequence; only
            # the events before the owner count as valid)
            keep = _map(out, lambda t: t.detach().clone())
            cached = ctx.cached
            if i < len(cached):
                cached[i] = keep
            elif i == len(cached):
                cached.append(keep)
            else:
                raise _SequenceChanged()
        return out

    def _learn_wiring(self, i, leaves, spec, prev):
        """Event i's arguments as references into event i-1's arguments / outputs (by identity),
        or nothing when a tensor among them was made by the model's glue."""
        prev_args, prev_out = prev
        by_out = {id(t): j for j, t in enumerate(prev_out) if torch.is_tensor(t)}
        by_arg = {id(t): j for j, t in enumerate(prev_args) if torch.is_tensor(t)}
        refs = []
        for leaf in leaves:
            if torch.is_tensor(leaf):
                if id(leaf) in by_out:
                    refs.append(("o", by_out[id(leaf)]))
                elif id(leaf) in by_arg:
                    refs.append(("a", by_arg[id(leaf)]))
                else:
                    return
            elif leaf is None or isinstance(leaf, (bool, int, float, str)):
                refs.append(("c", leaf))
            else:
                return
        self.wired[i] = (spec, tuple(refs))

    # ---- the loss closure ----------------------------------------------------------------------
    def __call__(self, model, samples, cuda_enabled):
        assert model is self.model
        if self.disabled:
            return self.loss_func(model, samples, cuda_enabled)
        try:
            return self._cached_call(model, samples, cuda_enabled)
        except _SequenceChanged:
            self._give_up()
            return self.loss_func(model, samples, cuda_enabled)

    def _give_up(self):
        # a forward that calls its cacheable modules in another order / number than the
        # recorded one (data-dependent control flow): the record cannot be trusted — drop it
        # for good and evaluate this and every later loss with plain full forwards
        import warnings
        warnings.warn("HookedPrefixLoss: the model's module call sequence is not fixed; "
                      "falling back to full forwards")
        self.disabled = True
        self.cache.clear()
        self.valid.clear()

    def _new_ctx(self, samples, lock):
        key = id(samples)
        self._touch(key, samples)
        limit = self._limit()
        cached = self.cache.setdefault(key, [])
        n_valid = min(self.valid.get(key, 0), limit)
        return _Ctx(self, key, n_valid, limit, cached, lock)

    def _finish_ctx(self, ctx):
        record_sequence = self.sequence is None
        if record_sequence:
            self.sequence = ctx.seen
            limit = self._limit()
        else:
            limit = ctx.limit
            if ctx.counter != len(self.sequence):
                raise _SequenceChanged()
        key = ctx.key
        self.valid[key] = min(max(self.valid.get(key, 0), limit), len(ctx.cached))
        del ctx.cached[self.valid[key]:]          # nothing recorded at / after the owner survives
        self.stats["events_total"] += ctx.counter
        self.stats["forwards"] += 1

    def _cached_call(self, model, samples, cuda_enabled):
        ctx = self._new_ctx(samples, lock=False)
        before = getattr(_TLS, "ctx", None)
        with self._patched():
            _TLS.ctx = ctx
            try:
                out = self.loss_func(model, samples, cuda_enabled)
            finally:
                _TLS.ctx = before
        self._finish_ctx(ctx)
        self.blen[ctx.key] = out[1]
        return out

    def _worker_profile(self, i):
        """ECOFLAP_LOCKSTEP_PROFILE=<file>: cProfile of worker 0's forwards (diagnosis only)."""
        path = os.environ.get("ECOFLAP_LOCKSTEP_PROFILE")
        if not path or i != 0:
            return None
        if getattr(self, "_prof", None) is None:
            import atexit
            import cProfile
            self._prof = cProfile.Profile()
            atexit.register(lambda: self._prof.dump_stats(path))
        return self._prof

    # ---- k evaluations in lock step ----------------------------------------------------------------
    def multi(self, model, items, cuda_enabled):
        """items: [(samples, theta_plus, theta_minus)] for up to pairs_in_flight() units ->
        [(loss(theta+), loss(theta-), batch_len)]."""
        evals = []
        for samples, tp, tm in items:
            evals += [(samples, tp), (samples, tm)]
        losses = None
        if not self.disabled and self.eval_batch > 1:
            try:
                try:
                    losses = self._lockstep(model, evals, cuda_enabled)
                except _SequenceChanged:
                    raise
                except Exception as ex:
                    if not self.lazy_slices:
                        raise
                    # glue that looks at a value it only seemed to pass on (a shape test between
                    # two blocks): hand every event out sliced per evaluation from here on
                    import warnings
                    warnings.warn(f"HookedPrefixLoss: lock step with shared intermediate values failed "
                                  f"({type(ex).__name__}: {ex}); slicing every event per evaluation")
                    self.lazy_slices = False
                    self.stats["lazy_slices_disabled"] = f"{type(ex).__name__}: {ex}"[:200]
                    losses = self._lockstep(model, evals, cuda_enabled)
            except _SequenceChanged:
                self._give_up()
                losses = None
        if losses is None:
            losses = self._sequential(model, evals, cuda_enabled)
        return [(losses[2 * i], losses[2 * i + 1], self._lens[id(items[i][0])]) for i in range(len(items))]

    def _sequential(self, model, evals, cuda_enabled):
        out = []
        param, home = self._pair_param, self._pair_home
        try:
            for samples, theta in evals:
                param.data = theta
                l, n = self(model, samples, cuda_enabled)
                self._lens[id(samples)] = n
                out.append(l.detach().clone())
        finally:
            param.data = home
        return out

    def _lockstep(self, model, evals, cuda_enabled):
        """-> the k losses, or None when this chunk has to run per evaluation (nothing recorded
        yet, unknown owner, batches of several shapes, an owner that is called twice, ...)."""
        from .prefix_cache import _family
        k = len(evals)
        if self.sequence is None:
            # the very first forward records the sequence, the wiring and every output
            samples, theta = evals[0]
            self._pair_param.data = theta
            try:
                self._cached_call(model, samples, cuda_enabled)
            finally:
                self._pair_param.data = self._pair_home
        owner_ev = self._limit()
        seq = self.sequence
        if (self.owner is None or k < 2 or seq[owner_ev] is not self.owner
                or sum(1 for m in seq if m is self.owner) != 1):
            return None
        fams = {_family(s) for s, _ in evals}
        if len(fams) != 1:
            return None
        fam = next(iter(fams))
        ctxs = [self._new_ctx(samples, lock=True) for samples, _ in evals]
        if len({c.n_valid for c in ctxs}) != 1:
            return None                           # (the per-evaluation path also brings them level)
        B = self.blen.get(ctxs[0].key)
        if B is None or any(self.blen.get(c.key) != B for c in ctxs):
            # a batch never seen before: one plain forward tells its length (and records it)
            return None
        thetas = [theta for _, theta in evals]
        self._poll_assumed()
        t0 = time.time()
        losses = self._run_lockstep(model, evals, ctxs, thetas, cuda_enabled, owner_ev, fam, B)
        self.stats["lockstep_seconds"] = self.stats.get("lockstep_seconds", 0.0) + time.time() - t0
        if losses is None:
            return None
        self.stats["lockstep_evals"] = self.stats.get("lockstep_evals", 0) + k
        # the guard: one evaluation per entry block (its slot rotating) also runs alone, through
        # the plain served forward, and must give the same loss bit for bit
        every = self.verify_batched == "all" or bool(os.environ.get("ECOFLAP_VERIFY_BATCHED"))
        if every or (fam, owner_ev) not in self._verified:
            self._verified.add((fam, owner_ev))
            sel = list(range(k)) if every else [self.stats.get("lockstep_checks", 0) % k]
            want = self._sequential(model, [evals[i] for i in sel], cuda_enabled)
            self.stats["lockstep_checks"] = self.stats.get("lockstep_checks", 0) + 1
            if self.defer_guard and not every and losses[sel[0]].is_cuda:
                self._guard_deferred([losses[i] for i in sel], want,
                                     f"the lock-step guard at {self.paths[self.owner]} ({self._pair_name})")
                return losses
            t1 = time.time()
            same = all(torch.equal(losses[i], w) for i, w in zip(sel, want))
            self.stats["host_blocked_seconds"] = self.stats.get("host_blocked_seconds", 0.0) + time.time() - t1
            self.check_assumed()      # (not finish_run: the forward patches stay on for the layer's next chunks)
            if not same and every and len(sel) == k and not os.environ.get("ECOFLAP_VERIFY_BATCHED"):
                # every chunk is being checked (small tensors: the probes cannot be trusted): the
                # chunk takes its per-evaluation losses, the next chunk is checked again
                self.stats["lockstep_corrected_chunks"] = self.stats.get("lockstep_corrected_chunks", 0) + 1
                return want
            if not same:
                import warnings
                self.stats["lockstep_disabled_at"] = self.paths[self.owner]
                self.stats["lockstep_mismatch"] = [[i, float(losses[i]), float(w)] for i, w in zip(sel, want)]
                warnings.warn("HookedPrefixLoss: a lock-step loss differs from the per-evaluation loss at "
                              f"{self.paths[self.owner]} ({self._pair_name}); batched evaluation is off "
                              "for the rest of the run")
                self.eval_batch = 1
                return None
        return losses

    # -- threads as coroutines: exactly one of {coordinator, workers} runs at any time ------------
    def _park(self, ctx):
        ctx.wake.clear()
        self._main_evt.set()                      # back to the coordinator
        ctx.wake.wait()
        if ctx.abort:
            raise _Abort()

    def _run_lockstep(self, model, evals, ctxs, thetas, cuda_enabled, owner_ev, fam, B):
        k = len(ctxs)
        self._main_evt = main_evt = threading.Event()
        stream = torch.cuda.current_stream() if cuda_enabled and torch.cuda.is_available() else None
        device = torch.cuda.current_device() if stream is not None else None
        grad = torch.is_grad_enabled()
        # persistent workers: a thread keeps its library handles, primitive caches and stream
        # state from chunk to chunk (a fresh thread pays for them again: ~4 ms per evaluation on
        # the CPU toy)
        while len(self._workers) < k:
            w = _Worker()
            w.start()
            self._workers.append(w)

        def make_body(i):
            ctx = ctxs[i]

            def body():
                try:
                    if ctx.abort:
                        return
                    _TLS.ctx = ctx
                    if device is not None:
                        torch.cuda.set_device(device)
                    with torch.set_grad_enabled(grad), (torch.cuda.stream(stream) if stream is not None
                                                        else contextlib.nullcontext()):
                        prof = self._worker_profile(i)
                        if prof is not None:
                            prof.enable()
                        try:
                            ctx.result = self.loss_func(model, evals[i][0], cuda_enabled)
                        finally:
                            if prof is not None:
                                prof.disable()
                except _Abort:
                    pass
                except BaseException as ex:       # noqa: BLE001 (re-raised by the coordinator)
                    ctx.error = ex
                finally:
                    _TLS.ctx = None
                    ctx.done = True
                    ctx.wake.clear()              # before the hand-over: the next job's start must not be lost
                    main_evt.set()
            return body

        for i, ctx in enumerate(ctxs):
            ctx.slot = i
            ctx.wake = self._workers[i].wake
            self._workers[i].job = make_body(i)

        def resume(ctx):
            main_evt.clear()
            ctx.wake.set()
            main_evt.wait()

        ok = False
        try:
            with self._patched():
                while True:
                    for ctx in ctxs:
                        if not ctx.done:
                            resume(ctx)
                            if ctx.error is not None:
                                raise ctx.error
                    live = [c for c in ctxs if not c.done]
                    if not live:
                        break
                    if len(live) != k or len({c.pending[0] for c in live}) != 1:
                        raise _SequenceChanged()
                    self._run_segment(live[0].pending[0], ctxs, thetas, owner_ev, fam, B)
                    for c in ctxs:
                        c.pending = None
            ok = True
        finally:
            if not ok:
                for ctx in ctxs:
                    ctx.abort = True
                    if not ctx.done:
                        resume(ctx)
        done_keys = set()
        for ctx in ctxs:
            if ctx.key not in done_keys:          # (theta+ and theta- of a unit share their batch)
                done_keys.add(ctx.key)
                self._finish_ctx(ctx)
            else:
                self.stats["events_total"] += ctx.counter
                self.stats["forwards"] += 1
        for c, (samples, _) in zip(ctxs, evals):
            self._lens[id(samples)] = c.result[1]
        return [c.result[0].detach().clone() for c in ctxs]

    # -- the coordinator's side: one segment for all parked evaluations ---------------------------
    # Arguments and outputs of an event travel in one of two forms: "cat" — ONE set of leaves whose
    # batch-leading tensors hold the k evaluations concatenated along dim 0 (everything else is
    # shared) — or "per": k sets of leaves.  A wired segment stays in cat form from event to
    # event: no slice, no re-concatenation, no per-evaluation Python between two blocks.
    def _run_segment(self, e0, ctxs, thetas, owner_ev, fam, B):
        k = len(ctxs)
        flat = [_flatten((c.pending[2], c.pending[3])) for c in ctxs]
        spec = flat[0][1]
        if any(f[1] != spec for f in flat) or any(c.pending[1] is not self.sequence[e0] for c in ctxs):
            raise _SequenceChanged()
        state = ctxs[0].pending[4]
        if self.autocast_at.get(e0, state) != state or any(c.pending[4] != state for c in ctxs):
            raise _SequenceChanged()
        per = [f[0] for f in flat]                # the heads' arguments come per evaluation
        cat = self._cat(per, B, (fam, e0))
        n = len(self.sequence)
        e = e0
        first_of = {}
        for i, c in enumerate(ctxs):
            first_of.setdefault(c.key, i)
        lazy = self.lazy_slices
        while True:
            mod = self.sequence[e]
            if cat is None and per is not None and e != e0:
                cat = self._cat(per, B, (fam, e))     # (behind an event that ran per evaluation)
            with torch.no_grad(), _autocast_as(self.autocast_at.get(e, state)):
                if e == owner_ev:
                    out_cat, out_per = self._owner_event(e, mod, spec, cat, per, thetas, fam, B, k)
                else:
                    out_cat, out_per = self._shared_event(e, mod, spec, cat, per, fam, B, k)
            nxt = e + 1
            w = self.wired.get(nxt) if nxt < n else None
            out_cat_leaves = _flatten(out_cat)[0] if out_cat is not None else None
            need_per = (out_cat is None or w is None or not lazy
                        or (e < ctxs[0].limit and not self.stats.get("verify_all_small_tensors")))
            if need_per and out_per is None:
                out_per = self._split(out_cat, k, B)
            # what the parked forwards get back for this event: inside a wired run the model only
            # hands the value on to the next (served) block, so every evaluation gets the SAME
            # un-sliced object; the last event of the run, which real glue consumes, is sliced
            for i, c in enumerate(ctxs):
                c.ready[e] = out_per[i] if need_per else out_cat
            if e < ctxs[0].limit and not self.stats.get("verify_all_small_tensors"):
                # an event before the owner that was not on record yet (the scored matrix moved
                # on): computed with the finished layers' weights, it is what every later
                # evaluation of these batches will be served.  (Not when the tensors are too small
                # for the invariance probes to be trusted: a shared pass that rounds differently
                # would then sit in the record and feed the checking evaluations too; the record
                # is left to those per-evaluation forwards.)
                for key, i in first_of.items():
                    cached = ctxs[i].cached
                    keep = _map(out_per[i], lambda t: t.detach().clone())
                    if e < len(cached):
                        cached[e] = keep
                    elif e == len(cached):
                        cached.append(keep)
                    else:
                        raise _SequenceChanged()
            if w is None:
                return
            # the next event's arguments by the recorded plumbing, in the form(s) at hand
            new_cat = new_per = None
            if out_cat is not None and cat is not None:
                new_cat = [out_cat_leaves[ref] if kind == "o" else cat[ref] if kind == "a" else ref
                           for kind, ref in w[1]]
            if need_per or new_cat is None:
                if per is None:
                    per = self._uncat(cat, k, B)
                out_per_leaves = [_flatten(o)[0] for o in out_per]
                new_per = [[out_per_leaves[i][ref] if kind == "o" else per[i][ref] if kind == "a" else ref
                            for kind, ref in w[1]] for i in range(k)]
            if need_per:
                for i, c in enumerate(ctxs):
                    a_ = _unflatten(w[0], new_per[i])[0]
                    c.expect[nxt] = a_[0] if a_ and torch.is_tensor(a_[0]) else None
            else:
                a_ = _unflatten(w[0], new_cat)[0]
                exp = a_[0] if a_ and torch.is_tensor(a_[0]) else None
                for c in ctxs:
                    c.expect[nxt] = exp
            cat, per, spec, e = new_cat, new_per, w[0], nxt

    @staticmethod
    def _uncat(cat, k, B):
        return [[t[i * B:(i + 1) * B] if (torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == k * B) else t
                 for t in cat] for i in range(k)]

    def _call(self, mod, spec, leaves):
        a, kw = _unflatten(spec, leaves)
        return self._real[mod](*a, **kw)

    def _cat(self, leaves, B, where):
        """k argument sets -> one: tensors whose leading dimension is the batch are concatenated
        along it, anything else must be the same for every evaluation.  A tensor that is not
        batch-leading but comes as k different objects (a position bias each forward rebuilds) is
        compared by value with a host sync ONCE per (family, event, position); after that its
        equality is assumed for the chunk at hand and verified on the device without a sync (one
        compare launch OR-ing into a flag that `finish_run` / the guard read): a violation is an
        error, never a silently different loss."""
        k = len(leaves)
        out = []
        for j in range(len(leaves[0])):
            col = [leaves[i][j] for i in range(k)]
            first = col[0]
            if torch.is_tensor(first):
                if all(t is first for t in col):
                    out.append(first)
                elif first.dim() > 0 and first.shape[0] == B and all(
                        torch.is_tensor(t) and t.shape == first.shape and t.dtype == first.dtype for t in col):
                    out.append(torch.cat(col, 0))
                elif all(torch.is_tensor(t) and t.shape == first.shape and t.dtype == first.dtype for t in col):
                    tag = where + (j,)
                    if tag in self._value_shared and first.device.type == "cuda":
                        from .prefix_cache import _differ_flag
                        a = [first.contiguous()] * (k - 1)
                        self._assumed = _differ_flag(a, [t.contiguous() for t in col[1:]], self._assumed)
                    elif all(torch.equal(t, first) for t in col[1:]):
                        self._value_shared.add(tag)
                    else:
                        return None
                    out.append(first)             # (a constant the glue rebuilds per forward)
                else:
                    return None
            else:
                if any(type(c) is not type(first) or c != first for c in col):
                    return None
                out.append(first)
        return out

    @staticmethod
    def _split(out, k, B):
        leaves, spec = _flatten(out)
        rows = []
        for i in range(k):
            rows.append(_unflatten(spec, [t[i * B:(i + 1) * B] if (torch.is_tensor(t) and t.dim() > 0
                                                                   and t.shape[0] == k * B) else t
                                          for t in leaves]))
        return rows

    def _slots_same(self, alone, outs):
        """every slot of a shared result == that evaluation's own result, bit for bit (one fused
        compare launch per slot OR-ing into one flag, ONE read-back)"""
        from .prefix_cache import _differ_flag
        flag = None
        t0 = time.time()
        for a, o in zip(alone, outs):
            fa = [t for t in _flatten(a)[0] if torch.is_tensor(t)]
            fb = [t for t in _flatten(o)[0] if torch.is_tensor(t)]
            if len(fa) != len(fb) or any(x.shape != y.shape or x.dtype != y.dtype for x, y in zip(fa, fb)):
                return False
            if sum(t.numel() for t in fa) < 65536 and not self.stats.get("verify_all_small_tensors"):
                # too few values for one comparison to rule out a lucky agreement (toy shapes: two
                # GEMM kernels that add 32 products in different orders round to the same fp16
                # value for most inputs — measured: one evaluation in 704 did not): from here on
                # every chunk is also evaluated per evaluation and compared
                self.verify_batched = "all"
                self.stats["verify_all_small_tensors"] = True
            flag = _differ_flag([x.contiguous() for x in fa], [y.contiguous() for y in fb], flag)
        same = flag is None or not bool(flag.item())
        self.stats["host_blocked_seconds"] = self.stats.get("host_blocked_seconds", 0.0) + time.time() - t0
        return same

    def _queue_same(self, alone, outs, what):
        """`_slots_same` without the read-back: the compare launches OR into the flag that
        `finish_run` (and the guard) read; a difference found there is an error."""
        from .prefix_cache import _differ_flag
        for a, o in zip(alone, outs):
            fa = [t for t in _flatten(a)[0] if torch.is_tensor(t)]
            fb = [t for t in _flatten(o)[0] if torch.is_tensor(t)]
            if len(fa) != len(fb) or any(x.shape != y.shape or x.dtype != y.dtype for x, y in zip(fa, fb)):
                raise RuntimeError(f"HookedPrefixLoss: per-slot owner of {what} returns another structure")
            if fa and fa[0].device.type != "cuda":
                if not all(torch.equal(x, y) for x, y in zip(fa, fb)):
                    raise RuntimeError(f"HookedPrefixLoss: per-slot owner of {what} differs from the "
                                       "per-evaluation call; rerun with eval_batch=1")
                continue
            self._assumed = _differ_flag([x.contiguous() for x in fa], [y.contiguous() for y in fb], self._assumed)
            self._assumed_what = what

    PAD_SLOTS = 2      # see prefix_cache.py: the library's fp32 GEMMs treat the LAST rows of a problem differently

    def _graphed_call(self, key, mod, spec, leaves):
        """`mod` on `leaves` through a HIP graph captured on first use (None: not capturable —
        the caller runs it eagerly).  The graph's inputs are static buffers; a leaf that IS the
        static output of the graph that ran just before (a wired run of blocks) is used in place,
        anything else is copied in (one fused copy launch)."""
        g = self._graphs.get(key)
        if g is False or not self.use_graphs:
            return None
        if not all(t.is_cuda for t in leaves if torch.is_tensor(t)):
            return None
        sig = tuple((tuple(t.shape), t.dtype) if torch.is_tensor(t) else ("c", t) for t in leaves)
        if g is not None and g["sig"] != sig:
            g = None
        if g is None:
            from .base_pruner import capture_graph
            static_in = [t if (torch.is_tensor(t) and getattr(t, "_ecoflap_static", False))
                         else (t.clone() if torch.is_tensor(t) else t) for t in leaves]
            graph = torch.cuda.CUDAGraph()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            try:
                with capture_graph(graph, pool=self._graph_pool, capture_error_mode="thread_local"):
                    out = self._call(mod, spec, static_in)
            except Exception as ex:               # the module does something a capture cannot hold
                self._graphs[key] = False
                self.stats.setdefault("not_capturable", []).append(
                    f"{self.paths[mod]}: {type(ex).__name__}: {ex}"[:200])
                torch.cuda.synchronize()
                return None
            for t in _flatten(out)[0]:
                if torch.is_tensor(t):
                    t._ecoflap_static = True      # (may be consumed in place by the next graph)
            g = self._graphs[key] = {"graph": graph, "in": static_in, "out": out, "sig": sig}
            self.stats["graph_captures"] = self.stats.get("graph_captures", 0) + 1
        pairs = [(d, s_) for d, s_ in zip(g["in"], leaves) if torch.is_tensor(d) and d is not s_]
        if pairs:
            from .prefix_cache import _flush_copies
            _flush_copies(pairs)
        g["graph"].replay()
        self.stats["graph_replays"] = self.stats.get("graph_replays", 0) + 1
        return g["out"]

    def _shared_call(self, mod, spec, cat, k, B, pad, key=None):
        """-> the event's output in cat form (k slots), computed at width k + pad"""
        if cat is None:
            return None
        try:
            if not pad:
                out = self._graphed_call(key, mod, spec, cat) if key is not None else None
                return out if out is not None else self._call(mod, spec, cat)
            wide = [torch.cat([t, t[(k - 1) * B:].repeat((pad,) + (1,) * (t.dim() - 1))], 0)
                    if (torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == k * B) else t for t in cat]
            out = self._graphed_call(key, mod, spec, wide) if key is not None else None
            if out is None:
                out = self._call(mod, spec, wide)
            leaves, ospec = _flatten(out)
            return _unflatten(ospec, [t[:k * B] if (torch.is_tensor(t) and t.dim() > 0
                                                    and t.shape[0] == (k + pad) * B) else t for t in leaves])
        except Exception as ex:                   # a module that cannot take the concatenated batch
            self.stats["shared_call_error"] = f"{self.paths[mod]}: {type(ex).__name__}: {ex}"[:300]
            return None

    def _shared_event(self, e, mod, spec, cat, per, fam, B, k):
        """One event for all k evaluations -> (output in cat form or None, per-evaluation outputs
        or None).  ONCE on the concatenated arguments when that gives every slot the bits the
        evaluation gets alone — measured on first use per (batch-shape family, width, event) on
        ALL slots (a GEMM library may treat the rows of one slot differently: hipBLASLt's fp32
        kernels do so with the last rows of a problem) — at width k or, failing that, at width
        k + PAD_SLOTS with the extra slots carrying a copy of the last evaluation and never read;
        otherwise k calls."""
        inv = self.invariant.get((fam, k, e))
        if inv is None:
            if per is None:
                per = self._uncat(cat, k, B)
            alone = [self._call(mod, spec, per[i]) for i in range(k)]
            self.stats["invariance_probes"] = self.stats.get("invariance_probes", 0) + 1
            inv, keep = False, None
            for pad in (0, self.PAD_SLOTS):
                out = self._shared_call(mod, spec, cat, k, B, pad)
                if out is not None and self._slots_same(alone, self._split(out, k, B)):
                    inv, keep = ("pad", pad), out
                    break
            self.invariant[(fam, k, e)] = inv
            if not inv:
                self.stats.setdefault("events_not_batch_invariant", []).append(self.paths[mod])
            elif inv[1]:
                self.stats.setdefault("events_shared_with_padding", []).append(self.paths[mod])
            return keep, alone
        if inv:
            out = self._shared_call(mod, spec, cat, k, B, inv[1], key=(fam, k, e, inv[1]))
            if out is not None:
                self.stats["events_shared"] = self.stats.get("events_shared", 0) + 1
                return out, None
        self.stats["events_per_evaluation"] = self.stats.get("events_per_evaluation", 0) + 1
        if per is None:
            per = self._uncat(cat, k, B)
        return None, [self._call(mod, spec, per[i]) for i in range(k)]

    def _owner_event(self, e, mod, spec, cat, per, thetas, fam, B, k):
        """The owning block of all k evaluations in ONE pass at batch k*B, the perturbed Linear
        applied per slot with its own theta (the very call an evaluation makes alone: same M, N,
        K, same kernel, same bits); checked against the per-evaluation call on first use of every
        matrix.  Falls back to k calls with theta re-pointed.  -> (cat-form output or None,
        per-evaluation outputs or None)."""
        import torch.nn.functional as F
        name = self._pair_name
        param, home = self._pair_param, self._pair_home
        lin = None
        if (name.endswith(".weight") and self._owner_ok.get((fam, name), True)
                and self._owner_block_ok.get((fam, e, k), True)):
            try:
                lin = self.model.get_submodule(name[:-len(".weight")])
            except AttributeError:
                lin = None
            if not isinstance(lin, torch.nn.Linear):
                lin = None
        if cat is not None and lin is not None:
            if getattr(lin, "_ecoflap_pinned", False):
                from ..shapes.fused import linear as _pinned, linear_or_torch as _linear
            else:
                _pinned, _linear = None, F.linear

            def per_slot(x):
                if x.shape[0] != k * B:           # the Linear's input must carry the k slots in front
                    raise _NotBatchLeading()
                if _pinned is not None and lin.bias is not None and lin.__dict__.get("_defer_now"):
                    ys = [_pinned(x[i * B:(i + 1) * B], thetas[i], None, library_bias=lin.bias) for i in range(k)]
                    if all(y is not None for y in ys):
                        lin._bias_pending = True
                        return torch.cat(ys, 0)
                if _pinned is not None:
                    lin._bias_pending = False
                bias = lin.bias if lin.bias is not None else lin.__dict__.get("_call_bias")
                return torch.cat([_linear(x[i * B:(i + 1) * B], thetas[i], bias) for i in range(k)], 0)

            had = "forward" in lin.__dict__
            old = lin.__dict__.get("forward")
            lin.forward = per_slot
            out = None
            try:
                out = self._call(mod, spec, cat)
            except _NotBatchLeading:
                out = None
            finally:
                if had:
                    lin.forward = old
                else:
                    del lin.forward
            if out is not None and (fam, name) not in self._owner_ok:
                # first use of this MATRIX's per-slot form: against the per-evaluation call.  The
                # first matrix of a block is checked on EVERY slot, with a host sync and a graceful
                # fallback (the block's other ops run at k*B too: this is what tells whether the
                # block is batch invariant at that width).  Once a block has passed, its other
                # matrices can only differ in the patched Linear — the same GEMM call an evaluation
                # makes alone — so their check (one rotating slot) is queued without a sync and
                # read at the guard / at the end of the run: a violation is an error there.
                # (tensors too small for a comparison to be trusted, `verify_all_small_tensors`:
                # every matrix is checked on every slot, with a sync and the graceful fallback)
                first_of_block = ((fam, e, k) not in self._owner_block_ok
                                  or bool(self.stats.get("verify_all_small_tensors")))
                picks = list(range(k)) if first_of_block else [self.stats.get("owner_checks", 0) % k]
                if per is None:
                    per = self._uncat(cat, k, B)
                alone = []
                try:
                    for i in picks:
                        param.data = thetas[i]
                        alone.append(self._call(mod, spec, per[i]))
                finally:
                    param.data = home
                self.stats["owner_checks"] = self.stats.get("owner_checks", 0) + 1
                slots = self._split(out, k, B)
                if first_of_block:
                    ok = self._slots_same(alone, [slots[i] for i in picks])
                    self._owner_ok[(fam, name)] = ok
                    self._owner_block_ok[(fam, e, k)] = ok and self._owner_block_ok.get((fam, e, k), True)
                else:
                    self._queue_same(alone, [slots[i] for i in picks], name)
                    self._owner_ok[(fam, name)] = True
            if out is None:
                self._owner_ok[(fam, name)] = False
            if self._owner_ok.get((fam, name), False):
                self.stats["owner_batched_evals"] = self.stats.get("owner_batched_evals", 0) + k
                return out, None
            self.stats.setdefault("owner_not_batchable", []).append(name)
        if per is None:
            per = self._uncat(cat, k, B)
        outs = []
        try:
            for i in range(k):
                param.data = thetas[i]
                outs.append(self._call(mod, spec, per[i]))
        finally:
            param.data = home
        return None, outs
