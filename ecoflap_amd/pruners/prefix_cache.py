"""Exact suffix-only re-forward for the zeroth-order loop.

The reference evaluates two FULL forwards per (layer, batch, noise) unit
(layer_single_base_pruner.py:530-536), although only one weight matrix differs from
the previous evaluation and everything upstream of the block that owns it is unchanged.
With 288 GB of HBM the activations at every stage boundary of every calibration batch
fit trivially (BLIP-2: 16 batches x 5.8 MB), so the loss closure below

  * keeps, per calibration batch, the state entering the stage (ViT block, T5 block, ...)
    that owns the matrix being scored, computed with the CURRENT weights — i.e. with the
    drifted weights every earlier layer was left with, exactly what a full forward would
    see, because layers are scored in parameter order and never change again;
  * evaluates each loss by running only the stages from there to the end.

In eval mode this is bit-identical to the full forward (the stage functions ARE the
model's forward: `forward` is their composition) and removes on average ~78 % of the
FLOPs of BLIP-2's scoring pass.  It is a drop-in `loss_func`: `(model, samples,
cuda_enabled) -> (loss, batch_len)`; `LayerSparsity` tells it which matrix is being
perturbed through the optional `begin_layer(name)` hook.
"""
import os
import time

import torch

from .base_pruner import capture_graph


def _sync_timed(stats):
    """torch.cuda.synchronize(), its wall time added to stats["host_blocked_seconds"]."""
    import time
    t0 = time.time()
    torch.cuda.synchronize()
    stats["host_blocked_seconds"] = stats.get("host_blocked_seconds", 0.0) + time.time() - t0


def _vision_language_result(state):
    return state["loss"]


def _vision_result(state):
    """-log softmax(logits / 100)[target], the reference's loss_vision (pruners/utils.py:47-67)."""
    logits = state["predictions"] / 100
    targets = state["targets"]
    probs = torch.nn.functional.softmax(logits, -1)
    idx = torch.arange(len(targets), device=targets.device)   # (no H2D copy: graph-capturable)
    return -probs[idx, targets].log().mean()


def _map_tensors(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    return obj


def _flush_copies(pairs):
    """All tensor copies of one state hand-over in ONE launch on the GPU (a state is 6-10 tensors;
    one small copy kernel each was 10 % of a FlanT5 matrix's step)."""
    if not pairs:
        return
    if pairs[0][0].device.type == "cuda" and len(pairs) > 1:
        from ..shapes import fused
        fused.multi_copy(pairs)
        return
    for d, s_ in pairs:
        d.copy_(s_, non_blocking=True)


def _differ_flag(fa, fb, flag=None):
    """int32[1] device flag, non-zero when the tensor lists differ in any BIT (one fused launch
    per 32 tensors instead of an eq + an and-reduce kernel per tensor; nothing is read back
    here).  Bitwise is the right notion for these checks: -0.0 vs 0.0 counts as different."""
    dev = fa[0].device if fa else torch.device("cpu")
    if flag is None:
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
    if len(fa) != len(fb):
        return flag.fill_(1)
    if dev.type != "cuda":           # host tensors (the CPU test doubles): plain torch
        if not all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(fa, fb)):
            flag.fill_(1)
        return flag
    from ..shapes import fused
    return fused.multi_compare(list(zip(fa, fb)), flag)


def _gather_copies(dst, src, out):
    if torch.is_tensor(dst):
        out.append((dst, src))
    elif isinstance(dst, dict):
        for k in dst:
            _gather_copies(dst[k], src[k], out)
    elif isinstance(dst, (list, tuple)):
        for d, s_ in zip(dst, src):
            _gather_copies(d, s_, out)


def _copy_tensors(dst, src):
    pairs = []
    _gather_copies(dst, src, pairs)
    _flush_copies(pairs)


def _is_batched(t, B):
    return torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == B


def _cat_states(states, B):
    """k per-evaluation states -> one state whose batch-leading tensors are concatenated along
    dim 0; everything else (position biases, python scalars) is shared by construction."""
    first = states[0]
    if torch.is_tensor(first):
        return torch.cat(states, 0) if _is_batched(first, B) else first.clone()
    if isinstance(first, dict):
        return {k: _cat_states([st[k] for st in states], B) for k in first}
    if isinstance(first, (list, tuple)):
        return type(first)(_cat_states([st[i] for st in states], B) for i in range(len(first)))
    return first


def _gather_slot(dst, src, i, B, out):
    if torch.is_tensor(dst):
        if _is_batched(src, B) and dst.shape[0] != src.shape[0]:
            out.append((dst[i * B:(i + 1) * B], src))
        elif i == 0:
            out.append((dst, src))
    elif isinstance(dst, dict):
        for k in dst:
            _gather_slot(dst[k], src[k], i, B, out)
    elif isinstance(dst, (list, tuple)):
        for d, s_ in zip(dst, src):
            _gather_slot(d, s_, i, B, out)


def _copy_slot(dst, src, i, B):
    """Write one evaluation's state into slot i of a concatenated static state."""
    pairs = []
    _gather_slot(dst, src, i, B, pairs)
    _flush_copies(pairs)


def _slice_state(state, i, B, k):
    if torch.is_tensor(state):
        if k > 1 and state.dim() > 0 and state.shape[0] == k * B:
            return state[i * B:(i + 1) * B]
        return state
    if isinstance(state, dict):
        return {a: _slice_state(b, i, B, k) for a, b in state.items()}
    if isinstance(state, (list, tuple)):
        return type(state)(_slice_state(b, i, B, k) for b in state)
    return state


def _first_slots(state, k, B, width):
    """The first k of `width` slots of a concatenated state (views)."""
    if torch.is_tensor(state):
        if state.dim() > 0 and state.shape[0] == width * B:
            return state[:k * B]
        return state
    if isinstance(state, dict):
        return {a: _first_slots(b, k, B, width) for a, b in state.items()}
    if isinstance(state, (list, tuple)):
        return type(state)(_first_slots(b, k, B, width) for b in state)
    return state


def _copy_first_slots(dst, src, k, B, width):
    _copy_tensors(dst, _first_slots(src, k, B, width))


def _family(samples):
    """Shape signature of a calibration batch: batches of one family replay the same graphs
    (BLIP-VQA batches differ in their number of answers: one chain of graphs per family)."""
    sig = []

    def walk(o, path):
        if torch.is_tensor(o):
            sig.append((path, tuple(o.shape), str(o.dtype)))
        elif isinstance(o, dict):
            for k in sorted(o, key=str):
                walk(o[k], path + (str(k),))
        elif isinstance(o, (list, tuple)):
            if o and all(isinstance(v, int) for v in o):
                # per-sample counts (VQA answers per question): stage 0 turns them into an index
                # tensor whose length is their sum — that, not the individual values, is shape
                sig.append((path, "counts", len(o), sum(o)))
                return
            sig.append((path, "seq", len(o)))
            for i, v in enumerate(o):
                walk(v, path + (i,))
        elif isinstance(o, (int, float, str, bool, type(None))):
            sig.append((path, o))
        else:
            sig.append((path, type(o).__name__))
    walk(samples, ())
    return tuple(sig)


class PrefixCachedLoss:
    """use_graphs=True (GPU only): the suffix from each entry stage is captured once into a
    HIP graph (torch.cuda.CUDAGraph -> hipGraph) and replayed for every later unit that
    re-enters there; the scoring loop is launch-bound (thousands of small kernels per
    forward), so replay removes the host from the critical path.  Graphs bake device
    addresses in, hence `requires_static_weights`: LayerSparsity then copies theta+/theta-
    into the parameter's own storage instead of re-pointing `param.data`."""

    def __init__(self, model, kind="vision_language", batch_len_fn=None, use_graphs=False,
                 two_lanes=False, n_lanes=None, eval_batch=1, verify_batched="entries",
                 group_batch=4, assume_not_invariant=(), pad_slots=2, batched_advance=True,
                 batch_owner=True):
        self.model = model
        self.use_graphs = bool(use_graphs)
        # eval_batch = k > 1 (graphs only): k evaluations of a layer (theta+/theta- of k/2 units)
        # share everything downstream of the block that owns the perturbed matrix, so that part
        # runs ONCE on their states concatenated along the batch dimension (see `_batched`)
        assert eval_batch >= 1 and (eval_batch == 1 or eval_batch % 2 == 0)
        self.eval_batch = int(eval_batch) if self.use_graphs else 1
        # "entries": the first chunk of every (entry stage, first shared stage) pair is also run
        # sequentially and compared bit for bit; "all": every chunk (tests on toy shapes, where a
        # whole-tensor comparison of a few hundred values can agree by luck); "first": 4 chunks
        assert verify_batched in ("first", "entries", "all")
        self.verify_batched = verify_batched
        # group_batch = g (1 < g < k): stages between the owning stage and the first shared one
        # that are batch invariant at g concatenated evaluations (measured: the EVA ViT-g blocks at
        # 4, not at 8 or 16) run once per GROUP of g evaluations instead of once per evaluation
        self.group_batch = int(group_batch) if (self.use_graphs and group_batch) else 0
        # stage-name prefixes treated as NOT shareable at k whatever the probe finds (always safe:
        # it only moves work from the shared pass to the per-evaluation / per-group part; the
        # tests use it to drive the group path on toy shapes, where everything is invariant)
        self.assume_not_invariant = tuple(assume_not_invariant)
        # pad_slots = p: a stage that is not batch invariant at k concatenated evaluations may be
        # so in its first k slots when p more slots follow them (measured: the fp32 Q-Former's
        # library GEMMs treat the last rows of a problem differently — only the last slot differs
        # at k = 4..24, the last two at 17, none at 18): the shared pass then runs at width k + p
        # from that stage on, the p extra slots carry a copy of some evaluation and are never read
        import os
        if os.environ.get("ECOFLAP_PAD_SLOTS") is not None:         # A/B and end-to-end checks
            pad_slots = int(os.environ["ECOFLAP_PAD_SLOTS"])
        self.pad_slots = int(pad_slots) if (self.use_graphs and pad_slots and self.eval_batch > 1) else 0
        import os
        if os.environ.get("ECOFLAP_BATCHED_ADVANCE") == "0":       # A/B and end-to-end checks
            batched_advance = False
        self.batched_advance = bool(batched_advance) and self.use_graphs and self.eval_batch > 1
        # batch_owner: the OWNING stage of the k evaluations also runs once, at batch k*B, with the
        # perturbed Linear alone applied per slot (one GEMM per evaluation on that slot's rows,
        # the same call as alone, its weight read straight from K1's scratch: no theta copies, no
        # per-evaluation graph replays); eager launches (the patched Linear differs per matrix).
        # Needs the shared pass to start right behind the owning stage; checked bit for bit
        # against the per-evaluation path on first use for every matrix (one rotating slot).
        if os.environ.get("ECOFLAP_BATCH_OWNER") == "0":           # A/B
            batch_owner = False
        self.batch_owner = bool(batch_owner) and self.use_graphs and self.eval_batch > 1
        self._owner_ok = {}         # (family, matrix name) -> True / False once checked
        self._owner_in = {}         # (family, entry, width) -> static concatenated input state
        self._fam_B = {}            # family -> batch length (learnt in `_batched`)
        self._adv_bad = set()       # (family, stage) whose batched advance once differed
        self._adv_pending = []      # queued bitwise checks of the batched advance
        self._rechecking = False    # inside the guard's second look at a mismatch
        # diagnostics (tools/diag/transient_hunt.py): a dict here receives clones of every stage
        # output of the batched path (per group / per evaluation / shared pass), keyed by where
        # they came from; None in production
        self.trace = None
        self.trace_subops = False   # with `trace`: also checksums of the EVA blocks' intermediates
        self._inject_mismatch_once = False
        self.gchains = {}           # (family, g) -> _StageGraphs at batch g*B (lane 0)
        self._group_ready = set()   # (lane id, family, entry, R, S) captured with the device quiescent
        self.bchains = {}           # k -> (_StageGraphs at batch k*B, tail graph, losses)
        # both guards are per batch-shape FAMILY: which library kernel a GEMM gets (Stream-K,
        # split-K) depends on M/N/K, so invariance measured at one shape says nothing about another
        self._verified = set()      # (family, entry, S) whose batched losses were checked bit for bit
        self.invariant = {}         # (family, k, stage) -> batch invariant on this system (probed)
        if n_lanes is None:
            n_lanes = 2 if two_lanes else 1
        assert n_lanes in (1, 2, 3, 4, 6, 8)
        self.n_lanes = n_lanes if self.use_graphs else 1
        self.two_lanes = self.n_lanes > 1
        self.extra_lanes = None     # replicas (model copy + stream + graphs), built lazily
        self.requires_static_weights = self.use_graphs
        self.chain = None           # per-stage graphs of lane A for the current batch family
        self.chains = {}            # family -> _StageGraphs
        self.families = {}          # id(samples) -> family
        self.max_families = 16      # beyond that, batches of new shapes replay eagerly
        self._warmed = False
        self.plan = model.stage_plan()
        self.result = _vision_result if kind == "vision" else _vision_language_result
        self.batch_len_fn = batch_len_fn
        self.entry = 0              # stage that owns the parameter being perturbed
        self.cache = {}             # id(batch) -> (stage index, state entering that stage)
        self.stats = {"stage_calls": 0, "stage_calls_full": 0, "advance_calls": 0,
                      "graph_captures": 0, "graph_replays": 0, "capture_seconds": 0.0}

    # ---- hook called by LayerSparsity before the units of a layer --------------------------
    def begin_layer(self, name):
        entry = self.stage_of(name)
        if entry > self.entry and self.batched_advance:
            self._advance_all(entry)
        self.entry = entry

    # ---- the cached prefix states of all batches, moved to a later stage together ----------
    def _advance_all(self, target):
        """Every cached batch state waits at the same stage; moving them to `target` one batch at
        a time is (batches x stages) latency-bound launches at batch B.  The stages on the way
        have already run as part of a shared pass — at width k (+ padding) or in groups of g — and
        were measured batch invariant there, so their captured graphs advance k (or g) batches
        per replay: slot i of the concatenated state carries exactly the bits batch i gets alone.
        One rotating slot per stage is also advanced alone and compared bit for bit; a stage
        without a usable graph, or one that ever differs, falls back to the per-batch path."""
        if not (self.use_graphs and self.eval_batch > 1 and self._warmed):
            return
        todo = {}
        for key, (idx, st) in self.cache.items():
            fam = self.families.get(key)
            B = self._fam_B.get(fam)
            if fam is None or B is None or not (0 < idx < target) or not _on_gpu(st):
                continue
            todo.setdefault((fam, idx, B), []).append(key)
        for (fam, idx, B), keys in todo.items():
            if len(keys) < 2:
                continue
            chain1 = self.chains.get(fam)
            states = [self.cache[k][1] for k in keys]
            reached = idx
            with torch.no_grad():
                for j in range(idx, target):
                    nxt = self._advance_stage(fam, j, states, B, chain1)
                    if nxt is None:
                        break
                    states, reached = nxt, j + 1
            if reached > idx:
                for k_, st in zip(keys, states):
                    self.cache[k_] = (reached, st)
                self.stats["advance_calls"] += (reached - idx) * len(keys)

    def _advance_stage(self, fam, j, states, B, chain1):
        k, p, g = self.eval_batch, self.pad_slots, self.group_batch
        if (fam, j) in self._adv_bad:
            return None
        plans = []        # (graphs, width, slots used)
        b = self.bchains.get((fam, k))
        if self.invariant.get((fam, k, j), False) and b and j in b[0].graphs:
            plans.append((b[0], k, k))
        b = self.bchains.get((fam, k + p)) if p else None
        if not plans and p and self.invariant.get((fam, k + p, j), False) and b and j in b[0].graphs:
            plans.append((b[0], k + p, k))
        gch = self.gchains.get((fam, g)) if g else None
        if not plans and g and self.invariant.get((fam, g, j), False) and gch and j in gch.graphs:
            plans.append((gch, g, g))
        if not plans:
            return None
        ch, width, used = plans[0]
        graph, static_in, static_out = ch.graphs[j]
        outs = []
        for a in range(0, len(states), used):
            chunk = states[a:a + used]
            for i, st in enumerate(chunk):
                _copy_slot(static_in, st, i, B)
            if j in ch.bridges:      # this stage's input is normally fed by a copy: we wrote it directly
                pass
            graph.replay()
            for i in range(len(chunk)):
                outs.append(_map_tensors(_slice_state(static_out, i, B, width), lambda t: t.clone()))
        self.stats["advance_batched_replays"] = (self.stats.get("advance_batched_replays", 0)
                                                 + (len(states) + used - 1) // used)
        # the check: one batch of this stage alone (captured chain if it has the stage, else eager)
        picks = [self.stats.get("advance_checks", 0) % len(states)]
        if self.verify_batched == "all" or os.environ.get("ECOFLAP_VERIFY_BATCHED"):
            picks = list(range(len(states)))          # every slot (A/B and end-to-end checks)
        for pick in picks:
            if chain1 is not None and j in chain1.graphs:
                alone = chain1.advance(j, j + 1, states[pick])
            else:
                alone = self.plan[j][2](states[pick])
            fa, fb = [], []
            _map_tensors(alone, lambda t: fa.append(t.clone()) or t)
            _map_tensors(outs[pick], lambda t: fb.append(t) or t)
            self.stats["advance_checks"] = self.stats.get("advance_checks", 0) + 1
            self._adv_pending.append((fam, j, fa, fb))
            if os.environ.get("ECOFLAP_DEBUG_BATCHED"):
                print(f"[advance] stage {self.plan[j][0]} slot {pick} of {len(states)} "
                      f"differs={bool(_differ_flag(fa, fb).item())}", flush=True)
        return outs

    def _settle_advance_checks(self):
        """Compare what `_advance_stage` queued (one host sync for all of them); called before the
        advanced states are first used."""
        pending, self._adv_pending = self._adv_pending, []
        bad = False
        import time
        t0 = time.time()
        # one fused bitwise-compare launch per check (shapes.fused.multi_compare), ONE read-back
        flags = [_differ_flag(fa, fb) for _, _, fa, fb in pending]
        verdict = torch.cat(flags).cpu().tolist() if flags else []
        for (fam, j, fa, fb), differs in zip(pending, verdict):
            if differs:
                self._adv_bad.add((fam, j))
                self.stats["advance_mismatch_at"] = self.plan[j][0]
                bad = True
        # (the first comparison waits for everything queued before it: host time blocked on the device)
        self.stats["host_blocked_seconds"] = (self.stats.get("host_blocked_seconds", 0.0)
                                              + time.time() - t0)
        if bad:
            # exactness first: throw the advanced states away, the per-batch path rebuilds them
            self.cache.clear()
            self.batched_advance = False

    def stage_of(self, name):
        for i, (_, prefixes, _) in enumerate(self.plan):
            for p in prefixes:
                if name.startswith(p):
                    return i
        return 0                    # unknown owner: full forward (always correct)

    def reset(self):
        self.cache.clear()
        self.families.clear()
        self.entry = 0

    # ---- the loss closure ----------------------------------------------------------------------
    def _ensure_cached(self, key, samples):
        if self._adv_pending:
            self._settle_advance_checks()
        idx, state = self.cache.get(key, (0, samples))
        if idx > self.entry:        # asked for an earlier stage than cached: start over
            idx, state = 0, samples
        if idx < self.entry:
            chain = getattr(self, "chain", None)
            if (chain is not None and idx > 0 and _on_gpu(state)
                    and all(j in chain.graphs for j in range(idx, self.entry))):
                state = chain.advance(idx, self.entry, state)     # already-captured stages
                self.stats["advance_calls"] += self.entry - idx
            else:
                with torch.no_grad():
                    for j in range(idx, self.entry):
                        state = self.plan[j][2](state)
                        self.stats["advance_calls"] += 1
            idx = self.entry
        self.cache[key] = (idx, state)

    def _account(self, idx):
        self.stats["stage_calls"] += len(self.plan) - idx
        self.stats["stage_calls_full"] += len(self.plan)

    def _batch_len(self, samples):
        if self.batch_len_fn is not None:
            return self.batch_len_fn(samples)
        if "text_input" in samples:
            return len(samples["text_input"])
        return len(samples["label"])

    def _use_family(self, samples):
        """Point `self.chain` (and every lane's) at the graphs of this batch's shape family;
        False when the family cap is reached (that batch then replays eagerly)."""
        key = id(samples)
        fam = self.families.get(key)
        if fam is None:
            fam = self.families[key] = _family(samples)
        if fam not in self.chains:
            if len(self.chains) >= self.max_families:
                self.chain = None
                return False
            self.chains[fam] = _StageGraphs(self, self.plan, stream=None)
        self.chain = self.chains[fam]
        self._fam = fam
        for lane in self.extra_lanes or []:
            lane.use_family(fam)
        return True

    def __call__(self, model, samples, cuda_enabled):
        assert model is self.model
        key = id(samples)
        graphable = self.use_graphs and self._use_family(samples)
        self._ensure_cached(key, samples)
        idx, state = self.cache[key]
        if graphable and idx > 0 and _on_gpu(state):
            loss = self._graphed_suffix(idx, state)
        else:
            loss = self.result(self._suffix(idx, state))
        self._account(idx)
        return loss, self._batch_len(samples)

    # ---- both evaluations of one unit at once (LayerSparsity uses it when present) -----------
    def supports_pairs(self):
        return self.n_lanes > 1 or self.eval_batch > 1

    def pairs_in_flight(self):
        if self.eval_batch > 1:
            return self.eval_batch // 2
        return max(1, self.n_lanes // 2)

    def begin_layer_weights(self, name, home):
        """Called once per layer: `home` is the parameter's own storage (lane 0)."""
        self._pair_name, self._pair_home = name, home
        if self.n_lanes > 1 and self.extra_lanes is None and home.device.type == "cuda":
            self.extra_lanes = [_Lane(self) for _ in range(self.n_lanes - 1)]

    def end_layer_weights(self, final):
        """Drifted weights of the finished layer go to every replica."""
        for lane in self.extra_lanes or []:
            lane.params[self._pair_name].data.copy_(final)

    def multi(self, model, items, cuda_enabled):
        """items: [(samples, theta_plus, theta_minus)] for up to pairs_in_flight() units ->
        [(loss(theta+), loss(theta-), batch_len)].  Evaluation 2i runs on lane 2i, 2i+1 on lane
        2i+1; lane 0 is the model itself on the current stream, the others are replicas on
        their own streams, all in flight together.  `join()` before reading the losses."""
        evals = []
        for samples, tp, tm in items:
            evals += [(samples, tp), (samples, tm)]
        if self.eval_batch > 1:
            losses = self._batched(model, evals, cuda_enabled)
            if losses is not None:
                return [(losses[2 * i], losses[2 * i + 1], self._batch_len(items[i][0]))
                        for i in range(len(items))]
            per_group = max(1, self.n_lanes // 2)
            if len(items) > per_group:       # nothing to share here: lane-sized groups instead
                out = []
                for g0 in range(0, len(items), per_group):
                    res = self.multi_lanes(model, items[g0:g0 + per_group], cuda_enabled)
                    self.join()
                    out += [(a.clone(), b.clone(), n) for a, b, n in res]
                return out
        return self.multi_lanes(model, items, cuda_enabled)

    def multi_lanes(self, model, items, cuda_enabled):
        evals = []
        for samples, tp, tm in items:
            evals += [(samples, tp), (samples, tm)]
        can_fork = (self.extra_lanes is not None and self._warmed
                    and len(evals) <= len(self.extra_lanes) + 1)
        states = []
        for samples, _ in evals:
            key = id(samples)
            if not (self.use_graphs and self._use_family(samples)):
                can_fork = False
            self._ensure_cached(key, samples)
            states.append(self.cache[key])
        if can_fork:
            can_fork = all(idx > 0 and _on_gpu(st) for idx, st in states)
        losses = [None] * len(evals)
        if not can_fork:
            for i, (samples, theta) in enumerate(evals):        # sequential on lane 0
                self._pair_home.copy_(theta)
                l, _ = self(model, samples, cuda_enabled)
                losses[i] = l.clone()
        else:
            for i in range(1, len(evals)):                      # replicas first, lane 0 last
                lane = self.extra_lanes[i - 1]
                self._use_family(evals[i][0])
                lane.params[self._pair_name].data.copy_(evals[i][1])
                losses[i] = lane.replay(*states[i])
                self._account(states[i][0])
            self._use_family(evals[0][0])
            self._pair_home.copy_(evals[0][1])
            losses[0] = self._graphed_suffix(*states[0])
            self._account(states[0][0])
        return [(losses[2 * i], losses[2 * i + 1], self._batch_len(items[i][0]))
                for i in range(len(items))]

    # ---- k evaluations, shared suffix run once --------------------------------------------
    def _sequential(self, model, evals, cuda_enabled):
        out = []
        for samples, theta in evals:
            self._pair_home.copy_(theta)
            l, _ = self(model, samples, cuda_enabled)
            out.append(l.clone())
        return out

    def _probe_invariance(self, entry, evals, states, B):
        """Which stages after `entry` give every slot of a batch-concatenated state exactly the
        bits they give that evaluation alone?  Arithmetic order is data independent, so one
        bitwise comparison of whole activation tensors per stage decides it for this shape.
        (Measured on MI355X / hipBLASLt: every FlanT5 stage is; the ViT-g blocks are not beyond
        4 concatenated evaluations — their fp16 GEMMs are Stream-K kernels, whose split of the
        K loop depends on the tile position — and the fp32 Q-Former is not in its last slot.)"""
        k, n = self.eval_batch, len(self.plan)
        with torch.no_grad():
            ins = []
            for (samples, theta), (_, st) in zip(evals, states):
                self._pair_home.copy_(theta)
                ins.append(self.plan[entry][2](st))
            while len(ins) < k:
                ins.append(ins[-1])
            g = self.group_batch if 1 < self.group_batch < k else 0

            def slots_equal(both, outs_, width):
                flat_a, flag = [], None
                for i, o in enumerate(outs_):
                    got = _slice_state(both, i, B, width)
                    flat_a, flat_b = [], []
                    _map_tensors(o, lambda t: flat_a.append(t) or t)
                    _map_tensors(got, lambda t: flat_b.append(t) or t)
                    flag = _differ_flag(flat_a, flat_b, flag if i else None)
                return not bool(flag.item()), flat_a

            for j in range(entry + 1, n - 1):
                outs = [self.plan[j][2](x) for x in ins]
                same, flat_a = slots_equal(self.plan[j][2](_cat_states(ins, B)), outs, k)
                if self.plan[j][0].startswith(self.assume_not_invariant or ("\0",)):
                    same = False
                self.invariant[(self._fam, k, j)] = same
                if self.pad_slots:       # the first k slots of a state of k + p
                    wide = self.plan[j][2](_cat_states(ins + [ins[-1]] * self.pad_slots, B))
                    same_p = slots_equal(wide, outs, k + self.pad_slots)[0]
                    if self.plan[j][0].startswith(self.assume_not_invariant or ("\0",)):
                        same_p = False
                    self.invariant[(self._fam, k + self.pad_slots, j)] = same_p
                    del wide
                if g:        # the same question for groups of g (first and last group of the chunk)
                    ok_g = all(slots_equal(self.plan[j][2](_cat_states(ins[a:a + g], B)),
                                           outs[a:a + g], g)[0] for a in (0, k - g))
                    self.invariant[(self._fam, g, j)] = ok_g
                if same and sum(t.numel() for t in flat_a) < 65536 and self.verify_batched != "all":
                    # too few values for one comparison to rule out a lucky agreement (toy
                    # shapes): fall back to checking every chunk against the sequential losses
                    self.verify_batched = "all"
                    self.stats["verify_all_small_tensors"] = True
                ins = outs
        self.stats["invariance_probes"] = self.stats.get("invariance_probes", 0) + 1
        self.stats["stages_not_batch_invariant"] = sorted(
            {self.plan[j][0] for (_, kk, j), ok in self.invariant.items()
             if not ok and kk == self.eval_batch})
        if self.pad_slots:
            self.stats["stages_shared_with_padding"] = sorted(
                {self.plan[j][0] for (_, kk, j), ok in self.invariant.items()
                 if ok and kk == self.eval_batch + self.pad_slots
                 and not self.invariant.get((self._fam, self.eval_batch, j), False)})
        self.stats["stages_invariant_in_groups"] = len(
            {j for (_, kk, j), ok in self.invariant.items() if ok and kk == self.group_batch
             and not self.invariant.get((self._fam, self.eval_batch, j), False)})

    def _batch_from(self, entry, evals, states, B):
        """(S, width, S_narrow): the shared pass starts at stage S > entry at `width` slots and
        continues from S_narrow at k slots: S_narrow..n-2 are batch invariant at k; S..S_narrow-1
        (possibly none: width == k, S == S_narrow) are so in the first k slots of k + pad_slots.
        (None, k, None): nothing to share."""
        n = len(self.plan)
        fam, k = self._fam, self.eval_batch
        if any((fam, k, j) not in self.invariant for j in range(entry + 1, n - 1)):
            self._probe_invariance(entry, evals, states, B)      # every new family is probed

        def first_shared(width):
            S = n - 1
            while S - 1 > entry and self.invariant.get((fam, width, S - 1), False):
                S -= 1
            return S if S <= n - 2 else None

        S_k = first_shared(k)
        if self.pad_slots and S_k is not None:
            # stages in front of the width-k part that are exact in the first k slots of k + p:
            # the pass starts there at width k + p and narrows to k slots at S_k
            S_p = S_k
            while S_p - 1 > entry and self.invariant.get((fam, k + self.pad_slots, S_p - 1), False):
                S_p -= 1
            if S_p < S_k:
                return S_p, k + self.pad_slots, S_k
        return S_k, k, S_k

    def _batched(self, model, evals, cuda_enabled):
        """losses of `evals` = [(samples, theta)] (all for the layer announced by begin_layer).
        Every evaluation of a layer shares all weights downstream of the block that owns the
        perturbed matrix.  Each evaluation runs alone (batch B graphs) from the owning stage up
        to stage S-1; its state goes into slot i of ONE state of batch k*B; stages S..n-2 run
        once on that state; the last stage (loss head) runs per slot, so each loss is reduced
        over its own batch exactly as alone.  S is the first stage from which everything is
        batch invariant on this system (`_probe_invariance`), so slot i of the shared part
        carries exactly the bits of the evaluation run alone; the first chunks are also checked
        end to end against the sequential losses.  Returns None when this chunk has to run
        sequentially (warm-up, nothing to share)."""
        k, n = self.eval_batch, len(self.plan)
        entry = self.entry
        if not self._warmed or entry < 1 or len(evals) > k:
            return None
        states = []
        fams = set()
        for samples, _ in evals:
            key = id(samples)
            if not self._use_family(samples):
                return None
            fams.add(self._fam)
            self._ensure_cached(key, samples)
            states.append(self.cache[key])
        if len(fams) != 1 or not all(idx == entry and _on_gpu(st) for idx, st in states):
            return None
        B = self._batch_len(evals[0][0])
        self._fam_B[self._fam] = B
        S, width, S_n = self._batch_from(entry, evals, states, B)
        if S is None:
            return None
        if width > k:
            self.stats["padded_shared_evals"] = self.stats.get("padded_shared_evals", 0) + len(evals)
        bundle = self.bchains.get((self._fam, width))
        if bundle is None:
            bundle = [_StageGraphs(self, self.plan, stream=None), None, None]
            self.bchains[(self._fam, width)] = bundle
        bchain = bundle[0]
        captured = S in bchain.graphs
        # 1. the per-evaluation part: owning stage (its theta in the parameter's storage), then
        #    the not-shareable stages, into the slots of the batched input
        #    (odd evaluations on the second lane — weight replica, own stream, own graphs — when
        #    there is one: this part is latency-bound at batch B, two streams fill the device)
        outs = []
        owner_done = False
        if (self.batch_owner and captured and S == entry + 1 and width == k and len(evals) == k
                and self._owner_ok.get((self._fam, self._pair_name), True)):
            owner_done = self._batched_owner(entry, S, evals, states, B, bchain)
        lanes = list(self.extra_lanes) if (self.extra_lanes and captured and not owner_done) else []
        main = torch.cuda.current_stream()
        for lane in lanes:
            lane.stream.wait_stream(main)        # K1's theta and the previous pass are complete
        # stages entry+1 .. R-1: batch invariant in groups of g evaluations (not at k)
        g = self.group_batch if 1 < self.group_batch < k else 0
        R = entry + 1
        if g and captured:
            while R < S and self.invariant.get((self._fam, g, R), False):
                R += 1
        used_groups = R > entry + 1 and not owner_done
        if owner_done:
            evals_iter = []
        elif used_groups:
            slot_in = bchain.graphs[S][1]
            todo = list(enumerate(zip(evals, states)))
            for gi, a in enumerate(range(0, len(todo), g)):
                lane = lanes[gi % (len(lanes) + 1) - 1] if (lanes and gi % (len(lanes) + 1)) else None
                items = [(i, theta, st) for i, ((_, theta), (_, st)) in todo[a:a + g]]
                self._run_group(lane, entry, R, S, items, slot_in, B)
            self.stats["grouped_evals"] = self.stats.get("grouped_evals", 0) + len(evals)
            evals_iter = []
        else:
            evals_iter = list(enumerate(zip(evals, states)))
        for i, ((samples, theta), (_, st)) in evals_iter:
            lane = lanes[i % (len(lanes) + 1) - 1] if (lanes and i % (len(lanes) + 1)) else None
            if lane is not None:
                out = lane.run_prefix(entry, S, st, self._pair_name, theta)
                with torch.cuda.stream(lane.stream):
                    if self.trace is not None:
                        self.trace[("pre", i)] = _map_tensors(out, lambda t: t.clone())
                    _copy_slot(bchain.graphs[S][1], out, i, B)
                continue
            self._pair_home.copy_(theta)
            out = self.chain.run_stage(entry, st)
            if S > entry + 1:
                out = self.chain.replay(entry + 1, out, stop=S)
            if self.trace is not None:
                self.trace[("pre", i)] = _map_tensors(out, lambda t: t.clone())
            if captured:
                _copy_slot(bchain.graphs[S][1], out, i, B)
            else:
                outs.append(_map_tensors(out, lambda t: t.clone()))
        for lane in lanes:
            main.wait_stream(lane.stream)
        if not captured:
            while len(outs) < width:
                outs.append(outs[-1])
            cat = _cat_states(outs, B)
            bchain.ensure(S, cat, stop=S_n if width > k else n - 1)
            _copy_tensors(bchain.graphs[S][1], cat)
        # 2. shared suffix, once: the padded stages at k + p slots, everything after them at k
        padded = width > k
        if padded:
            wide_out = bchain.replay(S, None, stop=S_n)
            if self.trace is not None:
                self.trace[("wide", width)] = {
                    j: _map_tensors(bchain.graphs[j][2], lambda t: t.clone())
                    for j in range(S, S_n)}
            bundle = self.bchains.get((self._fam, k))
            if bundle is None:
                bundle = [_StageGraphs(self, self.plan, stream=None), None, None]
                self.bchains[(self._fam, k)] = bundle
            narrow = bundle[0]
            if S_n not in narrow.graphs:
                narrow.ensure(S_n, _first_slots(wide_out, k, B, width), stop=n - 1)
            _copy_first_slots(narrow.graphs[S_n][1], wide_out, k, B, width)
            mid = narrow.replay(S_n, None, stop=n - 1)
            bchain, width = narrow, k
        else:
            mid = bchain.replay(S, None, stop=n - 1)
        # 3. loss head per slot (one graph for all slots)
        if bundle[1] is None:
            losses = torch.zeros(k, dtype=torch.float32, device=self._pair_home.device)
            with torch.no_grad():
                for i in range(k):      # eager once: library handles, workspaces
                    losses[i].copy_(self.result(self.plan[n - 1][2](_slice_state(mid, i, B, width))))
                graph = torch.cuda.CUDAGraph()
                with capture_graph(graph, pool=bchain.pool, capture_error_mode="thread_local"):
                    for i in range(k):
                        losses[i].copy_(self.result(self.plan[n - 1][2](_slice_state(mid, i, B, width))))
            bundle[1], bundle[2] = graph, losses
            self.stats["graph_captures"] += 1
        bundle[1].replay()
        if self.trace is not None:
            self.trace[("shared", width)] = {
                j: _map_tensors(bchain.graphs[j][2], lambda t: t.clone())
                for j in range(S_n if padded else S, n - 1) if j in bchain.graphs}
            self.trace["losses"] = bundle[2].clone()
        for _ in evals:
            self.stats["stage_calls"] += (S - entry) + (n - S) / k
            self.stats["stage_calls_full"] += n
        self.stats["batched_evals"] = self.stats.get("batched_evals", 0) + len(evals)
        losses = [bundle[2][i].clone() for i in range(len(evals))]
        import os
        check = not self._rechecking and (
                 self.verify_batched == "all" or bool(os.environ.get("ECOFLAP_VERIFY_BATCHED"))
                 or ((self._fam, entry, S) not in self._verified
                     and (self.verify_batched == "entries" or len(self._verified) < 4)))
        if check:
            self._verified.add((self._fam, entry, S))
            # "entries": ONE evaluation per check, its slot rotating from check to check (batch
            # invariance is a property of a slot position, not of a theta+/theta- pair; with the
            # GEMM library in its reproducible mode nothing else can make a slot differ);
            # "all" / "first": the whole chunk
            if self.verify_batched == "entries" and not os.environ.get("ECOFLAP_VERIFY_BATCHED"):
                sel = [self.stats.get("batched_checks", 0) % len(evals)]
            else:
                sel = list(range(len(evals)))
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            want = self._sequential(model, [evals[i] for i in sel], cuda_enabled)
            ev1.record()
            if self._inject_mismatch_once:       # tests: one loss of this check is off, once
                self._inject_mismatch_once = False
                losses[sel[0]] = losses[sel[0]] + 1.0
            import time
            t_chk = time.time()
            same = all(torch.equal(losses[i], w) for i, w in zip(sel, want))   # one sync
            # device time of the guard's own sequential evaluations (one-off per entry stage:
            # ~65 of them in a 588-matrix run)
            self.stats["guard_gpu_seconds"] = (self.stats.get("guard_gpu_seconds", 0.0)
                                               + ev0.elapsed_time(ev1) * 1e-3)
            self.stats["host_blocked_seconds"] = (self.stats.get("host_blocked_seconds", 0.0)
                                                  + time.time() - t_chk)
            self.stats["batched_checks"] = self.stats.get("batched_checks", 0) + 1
            if not same:
                if os.environ.get("ECOFLAP_DEBUG_BATCHED"):
                    print("batched mismatch at", self.plan[entry][0], self._pair_name, "S", S,
                          [float(losses[i]) for i in sel], [float(x) for x in want], flush=True)
                # Does it repeat?  Every-chunk verification of the whole BLIP-2 config (21 runs,
                # ~24 000 checked chunks) shows, in about one run in four, ONE slot of ONE chunk
                # off by ~7e-3 in the loss: ViT-g matrices only, either lane, any position of a
                # group, theta loaded by memcpy or by a kernel alike, every other chunk of the run
                # clean and the final table identical.  The library GEMMs are bit-reproducible call
                # to call (round 2, tools/diag/gemm_determinism.py in the history); what is left as a suspect sits below
                # this file (graph replays of the fp16 ViT-g block on two streams).  A one-off is
                # not a property of the batched path: only a mismatch that REPEATS switches a
                # feature off; the chunk's losses are the sequential ones either way.
                want2 = self._sequential(model, [evals[i] for i in sel], cuda_enabled)
                seq_stable = all(torch.equal(a, b_) for a, b_ in zip(want, want2))
                again = None
                if seq_stable:
                    self._rechecking = True
                    try:
                        again = self._batched(model, evals, cuda_enabled)
                    finally:
                        self._rechecking = False
                repeats = again is not None and not all(torch.equal(again[i], w)
                                                        for i, w in zip(sel, want))
                if not repeats:
                    self.stats.setdefault("transient_mismatches", []).append(
                        {"entry": self.plan[entry][0], "layer": self._pair_name,
                         "side": "sequential" if not seq_stable else "batched",
                         "slots": [[i, float(losses[i]), float(w), float(w2)]
                                   for i, w, w2 in zip(sel, want, want2)
                                   if not (torch.equal(losses[i], w) and torch.equal(w, w2))],
                         "grouped": bool(used_groups), "lanes": len(lanes) + 1})
                    # a mismatch that does not repeat = a forward of this run was not
                    # reproducible.  Its one known cause (the GEMM library's Stream-K hand-off,
                    # blas_guard.py) is switched off, so this is an error, not something to
                    # paper over with the sequential losses — unless the user chose to run in
                    # the library's default mode
                    if os.environ.get("ECOFLAP_ALLOW_STREAMK") != "1":
                        raise RuntimeError(
                            "non-reproducible loss evaluation (batched and sequential passes "
                            "differ once, then agree): "
                            f"{self.stats['transient_mismatches'][-1]}; "
                            "set ECOFLAP_ALLOW_STREAMK=1 to record it and carry on with the "
                            "sequential losses")
                    if len(sel) == len(evals):
                        return want2
                    return self._sequential(model, evals, cuda_enabled)
                if used_groups:
                    # the group path first: the shared pass has its own record of clean checks
                    self.stats["grouping_disabled_at"] = self.plan[entry][0]
                    self.group_batch = 0
                    self._verified.discard((self._fam, entry, S))     # re-check without groups
                    return self._sequential(model, evals, cuda_enabled)
                if padded:
                    # then the padded width: stages shared only thanks to the extra slots
                    self.stats["padding_disabled_at"] = self.plan[entry][0]
                    self.pad_slots = 0
                    self._verified.discard((self._fam, entry, S))
                    return self._sequential(model, evals, cuda_enabled)
                self.stats["batched_disabled_at"] = self.plan[entry][0]
                self.eval_batch = 1
                return self._sequential(model, evals, cuda_enabled)
        return losses

    def _batched_owner(self, entry, S, evals, states, B, bchain):
        """The owning stage of all k evaluations in ONE eager pass at batch k*B; the perturbed
        Linear runs per slot on its own rows with its own theta (K1's scratch, no copy).  -> True
        when the shared pass's input now holds the k results; False: the caller takes the
        per-evaluation path (first-use check failed, or the Linear cannot be patched)."""
        import torch.nn.functional as F
        from ..shapes.fused import linear as _pinned, linear_or_torch
        k = len(evals)
        name = self._pair_name
        if not name.endswith(".weight"):
            return False
        try:
            mod = self.model.get_submodule(name[:-len(".weight")])
        except AttributeError:
            return False
        if not isinstance(mod, torch.nn.Linear):
            return False
        _linear = linear_or_torch if getattr(mod, "_ecoflap_pinned", False) else F.linear
        fam = self._fam
        key = (fam, entry, k)
        cat_in = self._owner_in.get(key)
        if cat_in is None:
            self._owner_in.clear()               # one entry stage at a time: the previous one's buffers go
            cat_in = self._owner_in[key] = _cat_states([st for _, st in states], B)
        else:
            pairs = []
            for i, (_, st) in enumerate(states):
                _gather_slot(cat_in, st, i, B, pairs)
            _flush_copies(pairs)
        thetas = [theta for _, theta in evals]

        class _NotBatchLeading(Exception):
            pass

        def per_slot(x):
            if x.shape[0] != k * B:              # the Linear's input must carry the k slots in front
                raise _NotBatchLeading()
            # the module's own dispatch, slot by slot: the pinned hipBLASLt solution of this weight
            # shape where the shape modules use it (shapes/fused.py) — the very call an evaluation
            # makes alone —, with the bias left to the consuming op exactly when the module's own
            # forward would leave it there (`_defer_bias`): same kernel, same roundings, same bits
            if mod.bias is not None and mod.__dict__.get("_defer_now"):
                ys = [_pinned(x[i * B:(i + 1) * B], thetas[i], None, library_bias=mod.bias) for i in range(k)]
                if all(y is not None for y in ys):
                    mod._bias_pending = True
                    return torch.cat(ys, 0)
            mod._bias_pending = False
            # (`_call_bias`: EVA's qkv Linear is handed its q / v biases per call by the Attention)
            bias = mod.bias if mod.bias is not None else mod.__dict__.get("_call_bias")
            return torch.cat([_linear(x[i * B:(i + 1) * B], thetas[i], bias)
                              for i in range(k)], 0)

        had = "forward" in mod.__dict__
        old = mod.__dict__.get("forward")
        mod.forward = per_slot
        try:
            with torch.no_grad():
                out = self.plan[entry][2](cat_in)
        except _NotBatchLeading:
            self._owner_ok[(fam, name)] = False
            self.stats.setdefault("owner_not_batchable", []).append(name)
            return False
        finally:
            if had:
                mod.forward = old
            else:
                del mod.forward
        if (fam, name) not in self._owner_ok:
            # first use of this MATRIX's per-slot form: one slot against the per-evaluation path
            pick = self.stats.get("owner_checks", 0) % k
            self._pair_home.copy_(thetas[pick])
            alone = self.chain.run_stage(entry, states[pick][1])
            fa, fb = [], []
            _map_tensors(alone, lambda t: fa.append(t) or t)
            _map_tensors(_slice_state(out, pick, B, k), lambda t: fb.append(t) or t)
            t0 = time.time()             # (the read-back waits for everything queued so far)
            same = not bool(_differ_flag(fa, fb).item())
            self.stats["host_blocked_seconds"] = (self.stats.get("host_blocked_seconds", 0.0)
                                                  + time.time() - t0)
            self.stats["owner_checks"] = self.stats.get("owner_checks", 0) + 1
            self._owner_ok[(fam, name)] = same
            if not same:
                self.stats.setdefault("owner_not_batchable", []).append(name)
                return False
        _copy_tensors(bchain.graphs[S][1], out)
        self.stats["owner_batched_evals"] = self.stats.get("owner_batched_evals", 0) + k
        return True

    def _run_group(self, lane, entry, R, S, items, slot_in, B):
        """items = [(slot, theta, state)] (at most g): the owning stage per evaluation (its theta
        in the parameter's storage), stages entry+1..R-1 ONCE on the group's concatenated states
        (batch g*B graphs), stages R..S-1 per evaluation again (not invariant in groups: the
        Q-Former bridge), each result into its slot of the shared pass's input.  lane None = the
        model itself on the current stream; otherwise a replica on its own stream.  The first use
        of a (lane, entry, R, S) captures with the device quiescent."""
        g = self.group_batch
        fam = self._fam
        if lane is None:
            chain, home, stream = self.chain, self._pair_home, None
            gch = self.gchains.get((fam, g))
            if gch is None:
                gch = self.gchains[(fam, g)] = _StageGraphs(self, self.plan, stream=None)
            plan = self.plan
        else:
            chain, home, stream = lane.chain, lane.params[self._pair_name].data, lane.stream
            gch = lane.group_chain((fam, g))
            plan = lane.plan
        key = (id(lane), fam, entry, R, S)
        ready = key in self._group_ready

        def body():
            if not ready and (entry + 1) not in gch.graphs:
                # once per chain: eager pass at the group's batch (library handles, workspaces),
                # then the captures below
                outs = []
                for _, theta, st in items:
                    home.copy_(theta)
                    outs.append(_map_tensors(chain.run_stage(entry, st), lambda t: t.clone()))
                while len(outs) < g:
                    outs.append(outs[-1])
                cat = _cat_states(outs, B)
                with torch.no_grad():
                    x = cat
                    for j in range(entry + 1, R):
                        x = plan[j][2](x)
                gch.ensure(entry + 1, cat, stop=R)
            tr = self.trace
            for pos, (slot, theta, st) in enumerate(items):
                home.copy_(theta)
                own = chain.run_stage(entry, st)
                if tr is not None:
                    tr[("own", slot)] = _map_tensors(own, lambda t: t.clone())
                _copy_slot(gch.graphs[entry + 1][1], own, pos, B)
            mid = gch.replay(entry + 1, None, stop=R)
            if tr is not None:
                tr[("group", items[0][0])] = {
                    j: _map_tensors(gch.graphs[j][2], lambda t: t.clone())
                    for j in range(entry + 1, R)}
                tr.setdefault("_meta", {})[("group", items[0][0])] = (
                    gch, stream, entry, R, [slot for slot, _, _ in items])
                for j in range(entry + 1, R):
                    for name, t in gch.subops.get(j, {}).items():
                        v = t.reshape(-1, t.shape[-1]).view(torch.int16)
                        tr[("sub", items[0][0], j, name, "rows")] = v.sum(1, dtype=torch.int64)
                        tr[("sub", items[0][0], j, name, "cols")] = v.sum(0, dtype=torch.int64)
            for pos, (slot, _, _) in enumerate(items):
                x = _slice_state(mid, pos, B, g)
                if S > R:
                    x = chain.replay(R, x, stop=S)
                if tr is not None:
                    tr[("pre", slot)] = _map_tensors(x, lambda t: t.clone())
                _copy_slot(slot_in, x, slot, B)

        if ready:
            if stream is None:
                body()
            else:
                with torch.cuda.stream(stream):
                    body()
            return
        _sync_timed(self.stats)
        if stream is None:
            with torch.no_grad():
                body()
        else:
            with torch.cuda.stream(stream), torch.no_grad():
                body()
        _sync_timed(self.stats)
        self._group_ready.add(key)

    def pair(self, model, samples, cuda_enabled, theta_plus, theta_minus):
        return self.multi(model, [(samples, theta_plus, theta_minus)], cuda_enabled)[0]

    def join(self):
        for lane in self.extra_lanes or []:
            torch.cuda.current_stream().wait_stream(lane.stream)

    def _suffix(self, idx, state):
        out = state
        for j in range(idx, len(self.plan)):
            out = self.plan[j][2](out)
        return out

    def _graphed_suffix(self, idx, state):
        if not self._warmed:
            # the very first suffix runs eagerly: the warm-up torch asks for before any
            # capture (library handles, workspaces)
            self._warmed = True
            return self.result(self._suffix(idx, state))
        return self.chain.replay(idx, state)


class _StageGraphs:
    """One HIP graph per STAGE (ViT block, T5 block, ...), captured once and chained through
    static buffers: stage j's graph reads the tensors stage j-1's graph wrote.  A suffix from
    any entry stage is then a sequence of graph launches (~10 us each) and the total capture
    cost of a whole pass is one forward's worth, instead of one capture per entry stage."""

    def __init__(self, owner, plan, stream):
        self.owner = owner
        self.plan = plan
        self.stream = stream            # None: torch's capture side stream / current stream
        self.graphs = {}                # stage -> (graph, static_in, static_out)
        self.solo = {}                  # stage -> the same, captured standalone (run_stage)
        self.pool = None
        self.bridges = {}               # stage -> tensors to copy into its static input
        self.subops = {}                # stage -> {name: intermediate tensor} (diagnostics)

    def _capture(self, j, static_in):
        import time
        t0 = time.time()
        graph = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        # thread_local: RCCL's watchdog thread may query events while this thread captures
        kw = {"pool": self.pool, "capture_error_mode": "thread_local"}
        if self.stream is not None:
            kw["stream"] = self.stream
        last = (j == len(self.plan) - 1)
        from ..shapes import fused as _fused
        sink = {} if getattr(self.owner, "trace_subops", False) else None
        _fused.TRACE_SINK = sink
        try:
            with torch.no_grad(), capture_graph(graph, **kw):
                out = self.plan[j][2](static_in)
                if last:
                    out = {"__loss__": self.owner.result(out)}
        finally:
            _fused.TRACE_SINK = None
        if sink:
            self.subops[j] = sink
        self.graphs[j] = (graph, static_in, out)
        self.owner.stats["graph_captures"] += 1
        self.owner.stats["capture_seconds"] += time.time() - t0
        return out

    def run_stage(self, j, state):
        """Replay stage j alone on `state`; returns its static output (captured on first use)."""
        if j in self.graphs and j not in self.solo:      # already part of the chain: replay it alone
            _copy_tensors(self.graphs[j][1], state)
            self.graphs[j][0].replay()
            return self.graphs[j][2]
        if j not in self.solo:       # kept apart from the chain
            self._capture(j, _map_tensors(state, lambda t: t.clone()))
            self.solo[j] = self.graphs.pop(j)
        _copy_tensors(self.solo[j][1], state)
        self.solo[j][0].replay()
        return self.solo[j][2]

    def ensure(self, idx, state, stop=None):
        """Make sure stages idx..stop-1 are captured, chained output -> input (stages captured
        earlier from another entry are joined by a copy at replay: `bridges`)."""
        n = len(self.plan) if stop is None else stop
        prev_out = None
        for j in range(idx, n):
            if j in self.graphs:
                if prev_out is not None and self.graphs[j][1] is not prev_out:
                    self.bridges[j] = prev_out
                prev_out = self.graphs[j][2]
                continue
            static_in = prev_out if prev_out is not None else _map_tensors(state, lambda t: t.clone())
            prev_out = self._capture(j, static_in)

    def advance(self, idx, stop, state):
        """Run captured stages idx..stop-1 on `state`; returns a private copy of the state
        entering stage `stop` (the static buffers are overwritten by the next replay)."""
        _copy_tensors(self.graphs[idx][1], state)
        for j in range(idx, stop):
            if j != idx and j in self.bridges:
                _copy_tensors(self.graphs[j][1], self.bridges[j])
            self.graphs[j][0].replay()
        return _map_tensors(self.graphs[stop - 1][2], lambda t: t.clone())

    def replay(self, idx, state, stop=None):
        """state=None: the caller has already written the static input of stage idx."""
        n = len(self.plan) if stop is None else stop
        if state is not None:
            self.ensure(idx, state, stop)
            _copy_tensors(self.graphs[idx][1], state)
        for j in range(idx, n):
            if j != idx and j in self.bridges:
                _copy_tensors(self.graphs[j][1], self.bridges[j])
            self.graphs[j][0].replay()
        self.owner.stats["graph_replays"] += 1
        out = self.graphs[n - 1][2]
        return out["__loss__"] if stop is None else out


class _Lane:
    """Second evaluation lane: a full weight replica (7.4 GB for BLIP-2 — nothing next to
    288 GB) with its own graphs on its own stream, so the theta- suffix of a unit replays
    CONCURRENTLY with the theta+ suffix of lane A.  The suffixes are launch- and
    latency-bound (hundreds of ~5-15 us kernels on 128-384 token activations) and leave most
    CUs idle; two streams fill them.  Same kernels on the same bits -> same losses."""

    def __init__(self, owner):
        import copy
        self.owner = owner
        self.model = copy.deepcopy(owner.model)
        self.plan = self.model.stage_plan()
        self.params = dict(self.model.named_parameters())
        self.stream = torch.cuda.Stream()
        self.chains = {}
        self.chain = None
        self.fam = None
        self.warmed = False
        self.prepared = set()
        self.gchains = {}

    def group_chain(self, key):
        if key not in self.gchains:
            self.gchains[key] = _StageGraphs(self.owner, self.plan, stream=self.stream)
        return self.gchains[key]

    def use_family(self, fam):
        if fam not in self.chains:
            self.chains[fam] = _StageGraphs(self.owner, self.plan, stream=self.stream)
        self.chain, self.fam = self.chains[fam], fam

    def run_prefix(self, entry, S, state, name, theta):
        """On this lane's stream: theta into the replica's parameter, then stages entry..S-1 of
        the replica on `state`; returns the lane's static output state.  The first use of an
        (entry, S) pair captures its graphs with the device quiescent."""
        def go():
            self.params[name].data.copy_(theta)
            out = self.chain.run_stage(entry, state)
            if S > entry + 1:
                out = self.chain.replay(entry + 1, out, stop=S)
            return out
        if (self.fam, entry, S) in self.prepared:
            with torch.cuda.stream(self.stream):
                return go()
        _sync_timed(self.owner.stats)
        with torch.cuda.stream(self.stream), torch.no_grad():
            if not self.warmed:                    # once: eager pass on this stream
                self.warmed = True
                self.params[name].data.copy_(theta)
                out = state
                for j in range(entry, S):
                    out = self.plan[j][2](out)
                self.stream.synchronize()
            out = go()
        _sync_timed(self.owner.stats)
        self.prepared.add((self.fam, entry, S))
        return out

    def replay(self, idx, state):
        """Enqueue this lane's suffix on its stream; returns the static loss tensor."""
        if not self.warmed:                        # once: eager pass on this stream
            self.warmed = True
            torch.cuda.synchronize()
            with torch.cuda.stream(self.stream), torch.no_grad():
                out = state
                for j in range(idx, len(self.plan)):
                    out = self.plan[j][2](out)
                self.owner.result(out)
            self.stream.synchronize()
        self.chain.ensure(idx, state)              # captures (if any) before the fork
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)              # theta- copy and K1 are ordered on `main`
        with torch.cuda.stream(self.stream):
            loss = self.chain.replay(idx, state)
        return loss


def _on_gpu(state):
    found = []
    _map_tensors(state, lambda t: found.append(t.device.type) or t)
    return bool(found) and all(d == "cuda" for d in found)
