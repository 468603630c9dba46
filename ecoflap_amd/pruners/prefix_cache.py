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
import torch


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


def _copy_tensors(dst, src):
    if torch.is_tensor(dst):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_tensors(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s_ in zip(dst, src):
            _copy_tensors(d, s_)


class PrefixCachedLoss:
    """use_graphs=True (GPU only): the suffix from each entry stage is captured once into a
    HIP graph (torch.cuda.CUDAGraph -> hipGraph) and replayed for every later unit that
    re-enters there; the scoring loop is launch-bound (thousands of small kernels per
    forward), so replay removes the host from the critical path.  Graphs bake device
    addresses in, hence `requires_static_weights`: LayerSparsity then copies theta+/theta-
    into the parameter's own storage instead of re-pointing `param.data`."""

    def __init__(self, model, kind="vision_language", batch_len_fn=None, use_graphs=False,
                 two_lanes=False, n_lanes=None):
        self.model = model
        self.use_graphs = bool(use_graphs)
        if n_lanes is None:
            n_lanes = 2 if two_lanes else 1
        assert n_lanes in (1, 2, 4, 6, 8)
        self.n_lanes = n_lanes if self.use_graphs else 1
        self.two_lanes = self.n_lanes > 1
        self.extra_lanes = None     # replicas (model copy + stream + graphs), built lazily
        self.requires_static_weights = self.use_graphs
        self.chain = None           # per-stage graphs of lane A
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
        self.entry = self.stage_of(name)

    def stage_of(self, name):
        for i, (_, prefixes, _) in enumerate(self.plan):
            for p in prefixes:
                if name.startswith(p):
                    return i
        return 0                    # unknown owner: full forward (always correct)

    def reset(self):
        self.cache.clear()
        self.entry = 0

    # ---- the loss closure ----------------------------------------------------------------------
    def _ensure_cached(self, key, samples):
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

    def __call__(self, model, samples, cuda_enabled):
        assert model is self.model
        key = id(samples)
        self._ensure_cached(key, samples)
        idx, state = self.cache[key]
        if self.use_graphs and idx > 0 and _on_gpu(state):
            loss = self._graphed_suffix(idx, state)
        else:
            loss = self.result(self._suffix(idx, state))
        self._account(idx)
        return loss, self._batch_len(samples)

    # ---- both evaluations of one unit at once (LayerSparsity uses it when present) -----------
    def supports_pairs(self):
        return self.n_lanes > 1

    def pairs_in_flight(self):
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
        can_fork = self.extra_lanes is not None and self._warmed
        states = []
        for samples, _ in evals:
            key = id(samples)
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
                lane.params[self._pair_name].data.copy_(evals[i][1])
                losses[i] = lane.replay(*states[i])
                self._account(states[i][0])
            self._pair_home.copy_(evals[0][1])
            losses[0] = self._graphed_suffix(*states[0])
            self._account(states[0][0])
        return [(losses[2 * i], losses[2 * i + 1], self._batch_len(items[i][0]))
                for i in range(len(items))]

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
        if self.chain is None:
            self.chain = _StageGraphs(self, self.plan, stream=None)
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
        self.pool = None
        self.bridges = {}               # stage -> tensors to copy into its static input

    def _capture(self, j, static_in):
        import time
        t0 = time.time()
        graph = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        kw = {"pool": self.pool}
        if self.stream is not None:
            kw["stream"] = self.stream
        last = (j == len(self.plan) - 1)
        with torch.no_grad(), torch.cuda.graph(graph, **kw):
            out = self.plan[j][2](static_in)
            if last:
                out = {"__loss__": self.owner.result(out)}
        self.graphs[j] = (graph, static_in, out)
        self.owner.stats["graph_captures"] += 1
        self.owner.stats["capture_seconds"] += time.time() - t0
        return out

    def ensure(self, idx, state):
        """Make sure stages idx..end are captured, chained output -> input."""
        n = len(self.plan)
        if idx in self.graphs and all(j in self.graphs for j in range(idx, n)):
            return
        static_in = _map_tensors(state, lambda t: t.clone())
        for j in range(idx, n):
            if j in self.graphs:
                # joins an existing chain captured from a later entry: bridge by copy at replay
                self.bridges[j] = static_in
                break
            static_in = self._capture(j, static_in)

    def advance(self, idx, stop, state):
        """Run captured stages idx..stop-1 on `state`; returns a private copy of the state
        entering stage `stop` (the static buffers are overwritten by the next replay)."""
        _copy_tensors(self.graphs[idx][1], state)
        for j in range(idx, stop):
            if j != idx and j in self.bridges:
                _copy_tensors(self.graphs[j][1], self.bridges[j])
            self.graphs[j][0].replay()
        return _map_tensors(self.graphs[stop - 1][2], lambda t: t.clone())

    def replay(self, idx, state):
        self.ensure(idx, state)
        n = len(self.plan)
        _copy_tensors(self.graphs[idx][1], state)
        for j in range(idx, n):
            if j != idx and j in self.bridges:
                _copy_tensors(self.graphs[j][1], self.bridges[j])
            self.graphs[j][0].replay()
        self.owner.stats["graph_replays"] += 1
        return self.graphs[n - 1][2]["__loss__"]


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
        self.chain = _StageGraphs(owner, self.plan, stream=self.stream)
        self.warmed = False

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
