"""Host-side helpers of the zeroth-order loop and the stage-2 block loop that only run on the
GPU path in production (graph families, slot layout of batched evaluations, calibration
uniformity): their pure-tensor logic, on CPU."""
import numpy as np
import pytest
import torch

from ecoflap_amd.pruners import prefix_cache as PC
from ecoflap_amd.pruners.wanda import _uniform_calibration


def test_family_ignores_values_but_not_shapes():
    a = {"image": torch.zeros(8, 3, 4, 4), "text_input": torch.zeros(8, 5, dtype=torch.long)}
    b = {"image": torch.ones(8, 3, 4, 4), "text_input": torch.ones(8, 5, dtype=torch.long)}
    c = {"image": torch.ones(8, 3, 4, 4), "text_input": torch.ones(8, 6, dtype=torch.long)}
    assert PC._family(a) == PC._family(b) != PC._family(c)
    # VQA tuples: per-question answer counts matter through their sum only
    v1 = (torch.zeros(2, 3), torch.zeros(2, 4), torch.zeros(3, 5), torch.zeros(3), [1, 2])
    v2 = (torch.zeros(2, 3), torch.zeros(2, 4), torch.zeros(3, 5), torch.zeros(3), [2, 1])
    v3 = (torch.zeros(2, 3), torch.zeros(2, 4), torch.zeros(4, 5), torch.zeros(4), [2, 2])
    assert PC._family(v1) == PC._family(v2) != PC._family(v3)


def test_slot_layout_round_trip():
    B, k = 2, 4
    states = [{"x": torch.full((B, 3), float(i)), "mask": torch.full((B, 1, 1, 3), float(-i)),
               "bias": torch.arange(5.0).view(1, 5), "n": 7, "pair": [torch.full((B,), float(i))]}
              for i in range(k)]
    cat = PC._cat_states(states, B)
    assert cat["x"].shape == (k * B, 3) and cat["bias"].shape == (1, 5) and cat["n"] == 7
    for i in range(k):
        sl = PC._slice_state(cat, i, B, k)
        assert torch.equal(sl["x"], states[i]["x"]) and torch.equal(sl["mask"], states[i]["mask"])
        assert torch.equal(sl["pair"][0], states[i]["pair"][0])
        assert sl["bias"] is cat["bias"]
    # writing one evaluation into its slot leaves the others alone; shared tensors follow slot 0
    new = {"x": torch.full((B, 3), 9.0), "mask": torch.full((B, 1, 1, 3), 9.0),
           "bias": torch.zeros(1, 5), "n": 7, "pair": [torch.full((B,), 9.0)]}
    PC._copy_slot(cat, new, 2, B)
    assert torch.equal(PC._slice_state(cat, 2, B, k)["x"], new["x"])
    assert torch.equal(PC._slice_state(cat, 1, B, k)["x"], states[1]["x"])
    assert torch.equal(cat["bias"], torch.arange(5.0).view(1, 5))
    PC._copy_slot(cat, new, 0, B)
    assert torch.equal(cat["bias"], torch.zeros(1, 5))


def test_uniform_calibration_detection():
    inps = [torch.zeros(1, 4, 8) for _ in range(3)]
    caches = [{"attention_mask": torch.zeros(1, 1, 1, 4), "position_bias": None, "flag": False}
              for _ in range(3)]
    cpu_tensor_kwargs = _uniform_calibration(inps, caches, 3)
    assert cpu_tensor_kwargs is False            # kwargs tensors must live on the GPU to be replayed
    caches2 = [{"position_bias": None, "flag": False} for _ in range(3)]
    assert _uniform_calibration(inps, caches2, 3) is True
    ragged = [torch.zeros(1, 4, 8), torch.zeros(1, 5, 8), torch.zeros(1, 4, 8)]
    assert _uniform_calibration(ragged, caches2, 3) is False
    lists = [{"encoder_hidden_states": [torch.zeros(1, 2, 8)] * 2} for _ in range(3)]
    assert _uniform_calibration(inps, lists, 3) is False     # NLVR's twin states: eager path


class _CountingLoader:
    """Re-iterable loader that records how many batches were drawn and fails past a limit —
    stands in for a UPop entrypoint's full (shuffled) training loader."""

    def __init__(self, batches, limit):
        self.batches, self.limit, self.drawn = batches, limit, 0

    def __iter__(self):
        for i, b in enumerate(self.batches):
            assert i < self.limit, "the loader was consumed past the calibration prefix"
            self.drawn += 1
            yield b


def test_calibration_prefix_is_lazy_and_follows_the_reference_stop_rule():
    """layer_single_base_pruner.py:519-541: leave the loader once accum_samples >= num_samples;
    every noise draw spends samples.  Never list(loader)."""
    import torch
    from oracle_backend import OracleKernels
    from ecoflap_amd.pruners import LayerSparsity
    batches = [{"text_input": ["a"] * 4, "i": i} for i in range(100)]
    model = torch.nn.Linear(2, 2)
    for num_samples, num_noise, want in [(8, 1, 2), (9, 1, 3), (8, 2, 1), (16, 3, 2), (1, 1, 1)]:
        loader = _CountingLoader(batches, limit=want + 1)       # the break needs one look-ahead
        ls = LayerSparsity(model, loader, None, num_samples, 0.5, 0.6, "MEZO-GradOnly_sum", num_noise,
                           1e-3, {}, kernels=OracleKernels())
        got = ls.calibration_prefix()
        assert [b["i"] for b in got] == list(range(want)), (num_samples, num_noise)
        assert loader.drawn <= want + 1


def test_loss_table_keeps_the_loss_dtype():
    """(loss1 - loss2) / (2 eps) is evaluated in the loss tensor's own dtype (reference :544)."""
    import numpy as np
    import torch
    from oracle_backend import OracleKernels, torch_cpu_normal
    from ecoflap_amd.pruners import LayerSparsity
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 1)).eval()
    batches = [{"text_input": ["x"] * 4, "x": torch.randn(4, 8)} for _ in range(2)]

    def run(loss_dtype):
        def loss_fn(m, b, cuda_enabled):
            return (m(b["x"]).float().pow(2).mean() * 100).to(loss_dtype), 4
        torch.manual_seed(0)
        for p, q in zip(model.parameters(), init):
            p.data.copy_(q)
        np.random.seed(1)
        ls = LayerSparsity(model, batches, loss_fn, 8, 0.5, 0.6, "MEZO-GradOnly_sum", 1, 1e-3,
                           {"0.weight": "a", "1.weight": "b"}, kernels=OracleKernels(),
                           z_source=torch_cpu_normal, k1_form="single")
        ls.return_sparsity()
        return ls.loss_table, {k: float(v) for k, v in ls.importance_measure.items()}

    init = [p.data.clone() for p in model.parameters()]
    t32, s32 = run(torch.float32)
    tbf, sbf = run(torch.bfloat16)
    # the bf16 table holds bf16 values, and the projected gradient is the bf16 difference
    assert np.array_equal(tbf, torch.from_numpy(t32).to(torch.bfloat16).float().numpy())
    want = {}
    for name, rows in (("0.weight", (0, 1)), ("1.weight", (2, 3))):
        acc = np.float32(0)
        for r in rows:
            d = (torch.tensor(tbf[r, 0], dtype=torch.bfloat16) - torch.tensor(tbf[r, 1], dtype=torch.bfloat16)) / (2 * 1e-3)
            acc = np.float32(acc + np.float32(abs(float(d))))
        want[name] = float(acc)
    assert sbf == want and sbf != s32


def test_hooked_prefix_adapter_equals_full_forward_without_stage_plan():
    """A model WITHOUT stage_plan() (the reference's own modules swapped in):
    the hook adapter serves every block-list call before the owning block from its cache and the
    loss table, sparsity table and drifted weights equal the full-forward run bit for bit —
    through LayerSparsity directly and through the registered pruner; per evaluation
    (eval_batch = 1) and with the chunk's evaluations in lock step (round 5: the blocks behind
    the owner run ONCE on the evaluations concatenated, the owning Linear per slot)."""
    import numpy as np
    import torch
    from oracle_backend import OracleKernels, torch_cpu_normal
    from ecoflap_amd import load_pruner
    from ecoflap_amd.pruners import LayerSparsity
    from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss
    from ecoflap_amd.pruners.losses import loss_vision_language
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import Blip2T5, blip2_toy
    from ecoflap_amd.shapes.unstaged import hidden_stage_plan

    def run(mode):
        torch.manual_seed(4)
        model = blip2_toy().eval()
        assert not hasattr(model, "stage_plan")
        batches = S.image_text_batches(16, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6)
        mapping = {k: ".".join(k.split(".")[:4 if k.startswith("t5") else 3])
                   for k, v in model.named_parameters()
                   if v.dim() == 2 and ".block" in k and "relative_attention_bias" not in k}
        loss = loss_vision_language
        if mode != "full":
            loss = HookedPrefixLoss(model, loss_vision_language,
                                    ["visual_encoder.blocks", "t5_model.encoder.block",
                                     "t5_model.decoder.block"], ["ln_vision", "Qformer", "t5_proj"],
                                    eval_batch={"hooked": 1, "lock": 4, "lock_all": 6}[mode],
                                    verify_batched="all" if mode == "lock_all" else "entries")
        np.random.seed(42)
        ls = LayerSparsity(model, batches, loss, 16, 0.5, 0.6, "MEZO-GradOnly_sum", 1, 1e-3, mapping,
                           kernels=OracleKernels(), z_source=torch_cpu_normal)
        sp = ls.return_sparsity()
        # the forward patches are gone from the instances again (deepcopy / pickle see plain modules)
        assert not any("forward" in m.__dict__ for m in model.modules())
        if mode != "full":
            # a batch object the cache has not seen (same id or not) is never served another
            # batch's record: the cache holds the object it was filled from
            assert all(loss._held[k] is b for k, b in ((id(b), b) for b in batches))
            fresh = {k: v.clone() for k, v in batches[0].items()}
            with torch.no_grad():
                a = loss(model, fresh, False)[0]
                b = loss_vision_language(model, batches[0], False)[0]
            assert torch.equal(a, b)
            loss.close()
        return ls.loss_table, sp, {k: v.clone() for k, v in model.state_dict().items()}, loss

    def prune(prefix_cache, eval_batch=1):
        torch.manual_seed(4)
        model = blip2_toy().eval()
        batches = S.image_text_batches(8, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6)
        np.random.seed(42)
        cfg = dict(t5_prune_spec="2-0.5-1.0-1.0", vit_prune_spec="2-0.5-1.0-1.0",
                   t5_pruning_method="none", vit_pruning_method="none", num_samples=8,
                   max_sparsity_per_layer=0.6, num_data_first_stage=8,
                   sparsity_ratio_granularity="block", score_method="MEZO-GradOnly_sum",
                   kernels=OracleKernels(), z_source=torch_cpu_normal, eval_batch=eval_batch)
        pruner = load_pruner("blipt5_wanda_pruner", model, batches, cfg=cfg)
        pruner.prefix_cache = prefix_cache
        model, table = pruner.prune()
        return table, {k: v.clone() for k, v in model.state_dict().items()}, pruner.stage_stats

    with hidden_stage_plan(Blip2T5):
        full = run("full")
        for mode in ("hooked", "lock", "lock_all"):
            got = run(mode)
            assert np.array_equal(full[0], got[0]), mode
            assert full[1] == got[1], mode
            for k in full[2]:
                assert torch.equal(full[2][k], got[2][k]), (mode, k)
            st = got[3].stats
            assert st["events_served"] > 0.3 * st["events_total"], st  # most block calls came from cache
            if mode != "hooked":
                # every chunk but the first layer's (each batch's first forward runs alone: it
                # records the batch) went through the lock-step path, the owning block ran once
                # per chunk with the perturbed Linear per slot, nothing fell back for the run
                assert st["lockstep_evals"] >= 2 * len(full[0]) - 2 * 8 and st["owner_batched_evals"] > 0, st
                assert "lockstep_disabled_at" not in st and st["lockstep_checks"] >= 6, st
                assert st.get("events_shared", 0) + st.get("events_per_evaluation", 0) > 0
        t_full, w_full, _ = prune(False)
        for eb in (1, 4):
            t_hook, w_hook, stats = prune(True, eval_batch=eb)
            assert t_full == t_hook
            for k in w_full:
                assert torch.equal(w_full[k], w_hook[k]), k
            sf = stats["stage1"]["suffix_forward"]
            assert (sf.get("lockstep_evals", 0) > 0) == (eb > 1), sf
    assert hasattr(Blip2T5, "stage_plan")


def test_lock_step_evaluation_falls_back_where_it_cannot_share():
    """Lock step on models that resist it: (a) glue between the blocks that is NOT plumbing (each
    block's input is made by the model's own code) — every block becomes a segment of its own,
    losses unchanged; (b) a block argument that differs per evaluation without being
    batch-leading — the event runs per evaluation; (c) an owner that is called twice in a
    forward — the chunk runs per evaluation; (d) a call sequence that changes under lock step —
    the adapter gives up for good, losses still those of the plain forward."""
    import warnings
    import numpy as np
    import torch
    from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss

    class Net(torch.nn.Module):
        def __init__(self, glue, twice=False, odd=False):
            super().__init__()
            self.blocks = torch.nn.ModuleList(
                [torch.nn.Sequential(torch.nn.Linear(6, 6), torch.nn.Tanh()) for _ in range(4)])
            self.glue, self.twice, self.odd = glue, twice, odd

        def forward(self, batch):
            x = batch["x"]
            for i, blk in enumerate(self.blocks):
                if self.glue == "scale":
                    x = x * 1.25                                   # a new tensor per call: not plumbing
                x = blk(x)
                if self.twice and i == 1:
                    x = blk(x)
                if self.odd and i == 2 and batch.get("skip"):
                    break
            return {"loss": x.pow(2).mean()}

    def loss(m, b, c):
        return m(b)["loss"], b["x"].shape[0]

    def run(net_kw, eval_batch, batches):
        torch.manual_seed(0)
        model = Net(**net_kw).eval()
        name = "blocks.1.0.weight"
        param = dict(model.named_parameters())[name]
        home = param.data
        g = torch.Generator().manual_seed(1)
        thetas = [(home + 1e-2 * torch.randn(home.shape, generator=g)) for _ in range(2 * len(batches))]
        hooked = HookedPrefixLoss(model, loss, ["blocks"], eval_batch=eval_batch)
        hooked.begin_layer(name)
        hooked.begin_layer_weights(name, home)
        with torch.no_grad(), warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            items = [(b, thetas[2 * i], thetas[2 * i + 1]) for i, b in enumerate(batches)]
            got = []
            for rep in range(3):       # (the first call meets batches it has no record of)
                got = hooked.multi(model, items, False)
            want = []
            for b, tp, tm in items:
                pair = []
                for th in (tp, tm):
                    param.data = th
                    pair.append(loss(model, b, False)[0])
                param.data = home
                want.append(pair)
        hooked.end_layer_weights(None)
        assert param.data.data_ptr() == home.data_ptr()
        for (l1, l2, n), (w1, w2), (b, _, _) in zip(got, want, items):
            assert torch.equal(l1, w1) and torch.equal(l2, w2) and n == b["x"].shape[0]
        hooked.close()
        return hooked, w

    torch.manual_seed(3)
    batches = [{"x": torch.randn(3, 6)} for _ in range(2)]
    h, _ = run(dict(glue=None), 4, batches)
    assert h.stats["lockstep_evals"] == 8 and 2 in h.wired and 3 in h.wired
    h, _ = run(dict(glue="scale"), 4, batches)                     # (a)
    assert h.stats["lockstep_evals"] == 8 and not h.wired
    h, _ = run(dict(glue=None, twice=True), 4, batches)            # (c)
    assert h.stats.get("lockstep_evals", 0) == 0
    odd = [{"x": torch.randn(3, 6), "skip": False}, {"x": torch.randn(3, 6), "skip": True}]
    h, w = run(dict(glue=None, odd=True), 4, odd)                  # (d)
    assert h.disabled and any("not fixed" in str(x.message) for x in w)


def test_hooked_prefix_adapter_falls_back_when_the_call_sequence_changes():
    """A model whose forward calls its blocks a data-dependent number of times cannot be served
    from a recorded sequence: the adapter notices, warns once and evaluates plain full forwards
    (losses stay those of the full forward)."""
    import warnings
    import torch
    from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss

    class Odd(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([torch.nn.Linear(4, 4) for _ in range(3)])

        def forward(self, batch):
            x = batch["x"]
            for i in range(batch["depth"]):          # data-dependent control flow
                x = self.blocks[i](x)
            return {"loss": x.pow(2).mean()}

    torch.manual_seed(0)
    model = Odd().eval()
    loss = lambda m, b, c: (m(b)["loss"], 2)         # noqa: E731
    hooked = HookedPrefixLoss(model, loss, ["blocks"])
    b3 = {"x": torch.randn(2, 4), "depth": 3}
    b2 = {"x": torch.randn(2, 4), "depth": 2}
    hooked.begin_layer("blocks.2.weight")
    with torch.no_grad():
        assert torch.equal(hooked(model, b3, False)[0], loss(model, b3, False)[0])
        assert torch.equal(hooked(model, b3, False)[0], loss(model, b3, False)[0])    # served from the record
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert torch.equal(hooked(model, b2, False)[0], loss(model, b2, False)[0])
        assert hooked.disabled and len(w) == 1
        assert torch.equal(hooked(model, b3, False)[0], loss(model, b3, False)[0])
    assert all(m.forward.__func__ is torch.nn.Linear.forward for m in model.blocks)      # patches removed


def test_blas_guard_refuses_every_time_not_only_the_first(monkeypatch):
    """A failed probe is not cached: a second run in the same process (after the first refusal
    was caught) is refused again; the two properties — the variable in the environment,
    batch invariance — are reported apart, and a caller that never concatenates evaluations is
    not refused for missing batch invariance alone."""
    import pytest
    from ecoflap_amd import blas_guard as G
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(G, "_verified", {})
    monkeypatch.delenv("ECOFLAP_ALLOW_STREAMK", raising=False)
    calls = []

    def probe(result):
        def f(dev):
            calls.append(dev)
            return result
        return f

    # 1. the variable is set but the probe says "not batch invariant"
    monkeypatch.setenv(G.ENV, "1")
    monkeypatch.setattr(G, "_probe", probe((False, True)))
    for _ in range(2):
        with pytest.raises(RuntimeError, match="not batch invariant"):
            G.verify("cuda:0")
    assert len(calls) == 2 and G._verified == {}          # probed again, nothing cached
    with pytest.warns(UserWarning, match="could not be confirmed"):
        assert G.verify("cuda:0", need_batch_invariance=False) is True
    # 2. the variable is missing: refused on that alone, with the probe's result in the message
    monkeypatch.delenv(G.ENV)
    monkeypatch.setattr(G, "_probe", probe((True, True)))
    for _ in range(2):
        with pytest.raises(RuntimeError, match="is not in effect.*batch invariant at 16 evaluations = True"):
            G.verify("cuda:0", need_batch_invariance=False)
    assert G._verified == {}
    # 3. the override turns refusals into warnings, still every time
    monkeypatch.setenv("ECOFLAP_ALLOW_STREAMK", "1")
    for _ in range(2):
        with pytest.warns(UserWarning, match="is not in effect"):
            assert G.verify("cuda:0") is False
    # 4. a pass is cached
    monkeypatch.setenv(G.ENV, "1")
    n = len(calls)
    assert G.verify("cuda:0") is True and G.verify("cuda:0") is True
    assert len(calls) == n + 1 and G._verified == {0: (True, True)}


def test_hooked_prefix_cache_is_bounded_for_a_loader_of_fresh_batches():
    """A loader that yields NEW batch objects on every pass gets no reuse from an identity-keyed
    cache; it must not keep every batch (and its recorded activations) alive either."""
    from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([torch.nn.Linear(4, 4) for _ in range(3)])

        def forward(self, batch):
            x = batch["x"]
            for b in self.blocks:
                x = b(x)
            return {"loss": x.pow(2).mean()}

    torch.manual_seed(0)
    model = Net().eval()
    loss = lambda m, b, c: (m(b)["loss"], 2)         # noqa: E731
    hooked = HookedPrefixLoss(model, loss, ["blocks"], max_batches=8)
    hooked.begin_layer("blocks.2.weight")
    kept = [{"x": torch.randn(2, 4)} for _ in range(3)]
    with torch.no_grad():
        for i in range(40):
            fresh = {"x": torch.randn(2, 4)}
            assert torch.equal(hooked(model, fresh, False)[0], loss(model, fresh, False)[0])
            b = kept[i % 3]                            # a working set inside the bound stays served
            assert torch.equal(hooked(model, b, False)[0], loss(model, b, False)[0])
            assert len(hooked._held) <= 8 and len(hooked.cache) <= 8 and len(hooked.valid) <= 8
    assert hooked.stats["evicted"] >= 30
    assert hooked.stats["events_served"] >= 2 * 37     # blocks 0 and 1 of the kept batches, after their first pass


def test_k6_collector_refuses_inputs_written_in_place_before_the_flush():
    """The deferred one-launch column statistic keeps the hooked inputs by reference; an input
    modified in place before `flush()` is detected (tensor version counter) and refused loudly;
    `immediate=True` reduces inside the hook, as the reference does, and has no such limit."""
    import pytest
    from ecoflap_amd.pruners.wanda import _K6Collector

    class Kern:
        def __init__(self):
            self.seen = []

        def colsqnorm_accum_multi(self, items, ws=None):
            self.seen.append([x.clone() for _, x, *_ in items])

    class W:
        scaler_row, nsamples, n_dev = None, 0, None

    x = torch.ones(4, 3)
    col = _K6Collector(Kern())
    col.add(W(), x, 2)
    x.add_(1.0)                                       # in-place residual add after the Linear
    with pytest.raises(RuntimeError, match="modified in place"):
        col.flush()
    k = Kern()
    col = _K6Collector(k, immediate=True)
    w = W()
    x = torch.ones(4, 3)
    col.add(w, x, 2)                                  # reduced here
    x.add_(1.0)
    col.flush()
    col.end_sample()
    assert len(k.seen) == 1 and torch.equal(k.seen[0][0], torch.ones(4, 3)) and w.nsamples == 2
    # two launches in one sample (a Linear called twice): the private workspace is sized for
    # the larger one
    class K2(Kern):
        def colsqnorm_multi_workspace(self, items):
            return torch.zeros(sum(x.numel() for _, x, *_ in items), dtype=torch.uint8)
    col = _K6Collector(K2())
    w = W()
    col.add(w, torch.ones(8, 3), 2)
    col.add(w, torch.ones(2, 3), 2)                  # same Linear again: flushes the first
    col.flush()
    col.end_sample()
    assert col.private_workspace() and col.ws.numel() == 24


def test_pinned_linear_plumbing_on_the_cpu():
    """`pin_linears` (shapes/fused.py): the modules stay exactly nn.Linear (the reference's
    find_layers tests the type), a CPU forward is F.linear bit for bit, a deep copy's forward is
    bound to the COPY (the loop's lanes are deep copies with their own weights), the EVA block marks
    its biased Linears for the deferred-bias path and a forward leaves no bias pending."""
    import copy
    from ecoflap_amd.pruners.wanda import find_layers
    from ecoflap_amd.shapes import fused
    from ecoflap_amd.shapes.eva_vit import Block
    torch.manual_seed(0)
    blk = Block(32, 4, 64).eval()
    for p in blk.parameters():
        if p.dim() == 1:
            torch.nn.init.normal_(p, 0.0, 0.1)
    x = torch.randn(2, 5, 32)
    with torch.no_grad():
        want = blk(x, None)
    assert fused.pin_linears(blk) == 4 and fused.pin_linears(blk) == 0          # idempotent
    assert sorted(find_layers(blk)) == ["attn.proj", "attn.qkv", "mlp.fc1", "mlp.fc2"]
    assert all(type(m) is torch.nn.Linear for m in find_layers(blk).values())
    assert [bool(m.__dict__.get("_defer_bias")) for m in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2)] \
        == [False, True, True, True]
    with torch.no_grad():
        assert torch.equal(blk(x, None), want)
        assert fused.linear(x, blk.mlp.fc1.weight, blk.mlp.fc1.bias) is None       # CPU: the caller's F.linear
    assert not any(m.__dict__.get("_bias_pending") for m in blk.modules())
    twin = copy.deepcopy(blk)
    assert twin.mlp.fc1.forward.__self__ is twin.mlp.fc1
    with torch.no_grad():
        twin.mlp.fc1.weight.mul_(2.0)
        assert torch.equal(blk(x, None), want) and not torch.equal(twin(x, None), want)
    assert set(blk.state_dict()) == set(Block(32, 4, 64).state_dict())


def test_reference_style_forward_equals_the_staged_composition():
    """`Blip2T5.reference_forward` / `T5.reference_forward` — the forward written the way LAVIS
    writes it (one autocast region per tower, blip2_t5.py:116-168) and installed when the stage
    plan is hidden (shapes/unstaged.py) — gives the staged composition's loss and logits bit for
    bit, in fp32 and under (CPU) autocast at fp16 / bf16 weights."""
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import Blip2T5, blip2_toy
    from ecoflap_amd.shapes.t5 import T5, t5_config
    from ecoflap_amd.shapes.unstaged import hidden_stage_plan
    batches = S.image_text_batches(4, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6)
    for fp32 in (True, False):
        torch.manual_seed(0)
        m = blip2_toy(fp32=fp32).eval()
        m.cpu_autocast = not fp32
        with torch.no_grad():
            a, r = m(batches[0]), m.reference_forward(batches[0])
            assert torch.equal(a["loss"], r["loss"]) and torch.equal(a["logits"], r["logits"])
            with hidden_stage_plan(Blip2T5):
                assert not hasattr(m, "stage_plan")
                assert torch.equal(m(batches[1])["loss"], m.reference_forward(batches[1])["loss"])
            assert torch.equal(m(batches[1])["loss"], m.reference_forward(batches[1])["loss"])
    t = T5(t5_config(d_model=32, d_kv=8, num_heads=4, d_ff=64, num_layers=2, vocab_size=96), dtype=None,
           init_std=0.2).eval()
    tb = S.text_batches(4, 2, vocab=96, seed=3)
    with torch.no_grad():
        assert torch.equal(t(tb[0])["loss"], t.reference_forward(tb[0])["loss"])


@pytest.mark.parametrize("use_cache", [False, True])
def test_unstaged_path_on_a_huggingface_t5(use_cache):
    """The modules a reference user actually brings: `transformers`' own T5ForConditionalGeneration
    behind a LAVIS-shaped wrapper (`model(samples)["loss"]`, prunable prefix `t5_model`,
    LAVIS/lavis/models/t5_models/t5.py:60-90) — nothing of this build in the model.  Hooked per
    evaluation and in lock step == plain full forwards: loss table and sparsity table bit for bit.
    With `use_cache=False` the decoder's block calls are plumbing like the encoder's (tuple
    outputs: hidden states, position bias, cross-attention bias) and join the wired segments; with
    the library's default the decoder hands every block a KV-cache OBJECT, which the adapter
    cannot concatenate and does not try to: those calls run per evaluation, results unchanged."""
    transformers = pytest.importorskip("transformers")
    import warnings
    from oracle_backend import OracleKernels, torch_cpu_normal
    from ecoflap_amd.pruners import LayerSparsity
    from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss
    cfg = transformers.T5Config(vocab_size=96, d_model=32, d_kv=8, d_ff=64, num_layers=3, num_decoder_layers=3,
                                num_heads=4, feed_forward_proj="gated-gelu", dropout_rate=0.0,
                                tie_word_embeddings=False, decoder_start_token_id=0, pad_token_id=0)

    class Wrapper(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.t5_model = transformers.T5ForConditionalGeneration(cfg)

        def forward(self, samples):
            ids = samples["text_input"]
            out = self.t5_model(input_ids=ids, attention_mask=(ids != 0).long(), labels=samples["text_output"],
                                return_dict=True, use_cache=use_cache)
            return {"loss": out.loss}

    def loss_fn(m, b, c):
        return m(b)["loss"], len(b["text_input"])

    def run(mode):
        torch.manual_seed(0)
        model = Wrapper().eval()
        g = torch.Generator().manual_seed(1)
        batches = [{"text_input": torch.randint(1, 96, (2, 7), generator=g),
                    "text_output": torch.randint(1, 96, (2, 5), generator=g)} for _ in range(4)]
        mapping = {k: ".".join(k.split(".")[:4]) for k, v in model.named_parameters()
                   if v.dim() == 2 and ".block." in k and "relative_attention_bias" not in k}
        loss = loss_fn
        if mode != "full":
            loss = HookedPrefixLoss(model, loss_fn, ["t5_model.encoder.block", "t5_model.decoder.block"],
                                    eval_batch=1 if mode == "hooked" else 4)
        np.random.seed(3)
        ls = LayerSparsity(model, batches, loss, 8, 0.5, 0.6, "MEZO-GradOnly_sum", 1, 1e-3, mapping,
                           kernels=OracleKernels(), z_source=torch_cpu_normal)
        sp = ls.return_sparsity()
        if mode != "full":
            loss.close()
        return ls.loss_table, sp, loss

    full = run("full")
    for mode in ("hooked", "lock"):
        with warnings.catch_warnings():
            warnings.simplefilter("error")          # nothing gives up, nothing is switched off
            got = run(mode)
        assert np.array_equal(full[0], got[0]) and full[1] == got[1], mode
        h = got[2]
        assert len(h.sequence) == 6 and not h.disabled
        # encoder blocks 1, 2 always follow their predecessor by plumbing; decoder blocks 4, 5 do
        # when no cache object travels with them
        assert sorted(h.wired) == ([1, 2, 4, 5] if not use_cache else [1, 2]), sorted(h.wired)
        if mode == "lock":
            assert h.stats["lockstep_evals"] > 0 and h.stats.get("owner_batched_evals", 0) > 0


@pytest.mark.parametrize("name", ["t5_wanda_pruner", "vit_wanda_pruner"])
def test_unstaged_single_tower_pruners_equal_their_staged_runs(name):
    """`t5_wanda_pruner` (loss_language, `t5_model.{en,de}coder.block`) and `vit_wanda_pruner`
    (loss_vision through `predict()`, `visual.blocks`) on their shape modules with the stage plan
    hidden: the hook adapter finds the block lists, lock step included, and table and pruned
    weights equal the staged run's bit for bit."""
    from oracle_backend import OracleKernels, torch_cpu_normal
    from ecoflap_amd import load_pruner
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.eva_clip import EVACLIP, vit_toy
    from ecoflap_amd.shapes.t5 import T5, t5_config
    from ecoflap_amd.shapes.unstaged import hidden_stage_plan

    def run(eval_batch):
        torch.manual_seed(2)
        if name == "t5_wanda_pruner":
            model = T5(t5_config(d_model=32, d_kv=8, num_heads=4, d_ff=64, num_layers=2, vocab_size=96),
                       dtype=None, init_std=0.2).eval()
            batches = S.text_batches(8, 2, vocab=96, seed=5)
            cfg = dict(prune_spec="2-0.5-1.0-1.0", score_method="MEZO-GradOnly_avg")
        else:
            model = vit_toy().eval()
            batches = S.image_label_batches(8, 2, img_size=32, num_classes=5, seed=5)
            cfg = dict(prune_spec="3-0.5-1.0-1.0", score_method="MEZO-GradOnly_sum")
        np.random.seed(9)
        cfg.update(num_samples=8, num_data_first_stage=8, sparsity_ratio_granularity="block",
                   max_sparsity_per_layer=0.6, kernels=OracleKernels(), z_source=torch_cpu_normal,
                   eval_batch=eval_batch)
        pruner = load_pruner(name, model, batches, cfg=cfg)
        model, table = pruner.prune()
        return table, {k: v.clone() for k, v in model.state_dict().items()}, pruner.stage_stats

    staged = run(1)
    assert len(set(staged[0].values())) > 1
    with hidden_stage_plan(T5, EVACLIP):
        for eb in (1, 4):
            got = run(eb)
            assert got[0] == staged[0], eb
            for k in staged[1]:
                assert torch.equal(got[1][k], staged[1][k]), (eb, k)
            sf = got[2]["stage1"]["suffix_forward"]
            assert sf["events_served"] > 0 and (sf.get("lockstep_evals", 0) > 0) == (eb > 1), sf


def test_lock_step_shares_blocks_that_return_key_value_tuples_and_a_broadcast_bias():
    """The block interface of the T5 the reference vendors (LAVIS modeling_t5.py, old-HF style):
    a block returns `(hidden, (key, value), position_bias)`, the first block makes the
    position bias ([1, heads, S, S]: one for the whole batch) and every later block receives it,
    the stack keeps the key / value tuples.  Nested tuples of batch-leading tensors plus one tensor
    shared by value: the blocks behind the owner are wired and run once per chunk, losses equal
    the plain forwards bit for bit."""
    from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss

    class Block(torch.nn.Module):
        def __init__(self, first):
            super().__init__()
            self.q, self.k, self.v, self.o = (torch.nn.Linear(8, 8, bias=False) for _ in range(4))
            self.bias_table = torch.nn.Parameter(torch.randn(2, 5, 5) * 0.1) if first else None

        def forward(self, hidden, position_bias=None, use_cache=True):
            B, S, _ = hidden.shape
            sp = lambda t: t.view(B, S, 2, 4).transpose(1, 2)                       # noqa: E731
            q, k, v = sp(self.q(hidden)), sp(self.k(hidden)), sp(self.v(hidden))
            if position_bias is None:
                position_bias = self.bias_table.unsqueeze(0)                            # [1, heads, S, S]
            a = torch.softmax(q @ k.transpose(-1, -2) + position_bias, -1) @ v
            hidden = hidden + self.o(a.transpose(1, 2).reshape(B, S, 8))
            return (hidden, (k, v) if use_cache else None, position_bias)

    class Stack(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.block = torch.nn.ModuleList([Block(i == 0) for i in range(4)])
            self.head = torch.nn.Linear(8, 3)

        def forward(self, batch):
            hidden, bias, present = batch["x"], None, ()
            for blk in self.block:
                hidden, kv, bias = blk(hidden, position_bias=bias, use_cache=True)
                present = present + (kv,)
            return {"loss": torch.nn.functional.cross_entropy(self.head(hidden).flatten(0, 1), batch["y"].flatten()),
                    "past": present}

    def loss(m, b, c):
        return m(b)["loss"], b["x"].shape[0]

    torch.manual_seed(0)
    model = Stack().eval()
    name = "block.1.k.weight"
    param = dict(model.named_parameters())[name]
    home = param.data
    g = torch.Generator().manual_seed(1)
    batches = [{"x": torch.randn(3, 5, 8, generator=g), "y": torch.randint(0, 3, (3, 5), generator=g)}
               for _ in range(2)]
    thetas = [home + 1e-2 * torch.randn(home.shape, generator=g) for _ in range(4)]
    hooked = HookedPrefixLoss(model, loss, ["block"], eval_batch=4)
    hooked.begin_layer(name)
    hooked.begin_layer_weights(name, home)
    items = [(b, thetas[2 * i], thetas[2 * i + 1]) for i, b in enumerate(batches)]
    with torch.no_grad():
        for rep in range(3):
            got = hooked.multi(model, items, False)
        want = []
        for b, tp, tm in items:
            pair = []
            for th in (tp, tm):
                param.data = th
                pair.append(loss(model, b, False)[0])
            param.data = home
            want.append(pair)
    hooked.end_layer_weights(None)
    for (l1, l2, n), (w1, w2) in zip(got, want):
        assert torch.equal(l1, w1) and torch.equal(l2, w2) and n == 3
    assert hooked.stats["lockstep_evals"] == 8 and {2, 3} <= set(hooked.wired), (hooked.stats, sorted(hooked.wired))
    assert hooked.stats["events_shared"] > 0
    hooked.close()


def test_lock_step_hands_an_error_of_the_models_own_code_to_the_caller_and_recovers():
    """An exception raised by the model's forward in the middle of a lock-step chunk reaches the
    caller of `multi()` (after the one retry with per-evaluation slices that a failure inside shared
    values earns), leaves no worker stuck, and the adapter scores the same chunk correctly
    afterwards; `close()` ends every worker thread."""
    import threading
    import warnings
    from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList(
                [torch.nn.Sequential(torch.nn.Linear(6, 6), torch.nn.Tanh()) for _ in range(4)])
            self.calls, self.bomb = 0, None

        def forward(self, batch):
            x = batch["x"]
            for blk in self.blocks:
                x = blk(x)
            self.calls += 1
            if self.bomb is not None and self.calls >= self.bomb:
                raise ValueError("user code failed")
            return {"loss": x.pow(2).mean()}

    def loss(m, b, c):
        return m(b)["loss"], b["x"].shape[0]

    before = {t.ident for t in threading.enumerate()}
    torch.manual_seed(0)
    model = Net().eval()
    name = "blocks.1.0.weight"
    param = dict(model.named_parameters())[name]
    home = param.data
    g = torch.Generator().manual_seed(1)
    batches = [{"x": torch.randn(3, 6, generator=g)} for _ in range(2)]
    thetas = [home + 1e-2 * torch.randn(home.shape, generator=g) for _ in range(4)]
    hooked = HookedPrefixLoss(model, loss, ["blocks"], eval_batch=4)
    hooked.begin_layer(name)
    hooked.begin_layer_weights(name, home)
    items = [(b, thetas[2 * i], thetas[2 * i + 1]) for i, b in enumerate(batches)]
    with torch.no_grad(), warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        hooked.multi(model, items, False)
        good = hooked.multi(model, items, False)
        model.bomb = model.calls + 3
        with pytest.raises(ValueError, match="user code failed"):
            hooked.multi(model, items, False)
        model.bomb = None
        hooked.abort_run()
        again = hooked.multi(model, items, False)
    for (a1, a2, _), (b1, b2, _) in zip(good, again):
        assert torch.equal(a1, b1) and torch.equal(a2, b2)
    hooked.end_layer_weights(None)
    assert param.data.data_ptr() == home.data_ptr()
    hooked.close()
    import time
    time.sleep(0.3)
    assert {t.ident for t in threading.enumerate()} <= before


def test_evaluations_per_pass_are_sized_from_the_calibration_set():
    """eval_batch=0 (the default): all of a layer's evaluations in one pass up to ~256 samples,
    powers of two in [16, 64]; an explicit value is taken as it is."""
    import torch
    from ecoflap_amd.pruners.wanda import BLIPT5LayerWandaPruner as BLIPT5WandaPruner

    def rule(batch, n_first, noise=1, explicit=0):
        p = BLIPT5WandaPruner.__new__(BLIPT5WandaPruner)
        p.eval_batch, p.num_data_first_stage, p.num_noise, p.process_group = explicit, n_first, noise, None
        p.data_loader = [{"image": torch.zeros(batch, 3, 2, 2), "text_input": torch.zeros(batch, 3)}]
        return p._eval_batch()

    assert rule(8, 128) == 32          # config 3: 16 units -> 32 evaluations of 8 pairs
    assert rule(8, 1024) == 32         # configs[3]: capped at 256 samples per pass
    assert rule(1, 32) == 64           # the launchers' batch size 1, 32 samples: one pass per layer
    assert rule(1, 128) == 64          # capped at 64 evaluations
    assert rule(8, 8) == 16 and rule(2, 8) == 16        # toy sets: the floor
    assert rule(4, 64, noise=2) == 64
    assert rule(8, 128, explicit=4) == 4 and rule(1, 32, explicit=1) == 1
    p = BLIPT5WandaPruner.__new__(BLIPT5WandaPruner)
    p.eval_batch, p.num_data_first_stage, p.num_noise, p.process_group = 0, 128, 1, None
    p.data_loader = iter([])           # not indexable: the old default
    assert p._eval_batch() == 16
