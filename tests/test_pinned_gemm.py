"""The forward's 16-bit Linears through hipBLASLt with the solution pinned per weight shape
(csrc/gemm_pinned.hip, shapes/fused.py): for every Linear shape of BLIP-2 the chosen solution is
not a Stream-K / split-K kernel, equals torch's GEMM to rounding, gives the SAME BITS for a row
whatever travels with it (1, 8 x 257 or 16 x 8 x 257 rows; first, middle, last slot) and on every
repeated call — with TENSILE_STREAMK_DATA_PARALLEL removed from the environment of a child
process too (the property must not hang on that variable)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [  # (N, K, dtype, bias, rows of one evaluation)
    (4224, 1408, torch.float16, False, 8 * 257), (1408, 1408, torch.float16, True, 8 * 257),
    (6144, 1408, torch.float16, True, 8 * 257), (1408, 6144, torch.float16, True, 8 * 257),
    (6144, 1408, torch.float16, True, 257),            # batch size 1: the launchers' default
    (2048, 2048, torch.bfloat16, False, 8 * 48), (5120, 2048, torch.bfloat16, False, 8 * 48),
    (2048, 5120, torch.bfloat16, False, 8 * 48), (32128, 2048, torch.bfloat16, False, 8 * 16),
    # the fp32 Q-Former (32 queries per image; cross-attention keys from 257 image tokens)
    (768, 768, torch.float32, True, 8 * 32), (3072, 768, torch.float32, True, 8 * 32),
    (768, 3072, torch.float32, True, 32), (768, 1408, torch.float32, True, 8 * 257),
]


def _check_all():
    # every shape through its pinned solution, fast or not — for THIS function only: until round 6
    # the variable stayed set for the rest of the pytest process and every later run in it (a
    # whole config 3 included) took the pinned solutions, ending with another table than a fresh
    # process does (tests/test_full_configs.py)
    before = os.environ.get("ECOFLAP_PINNED_GEMM")
    os.environ["ECOFLAP_PINNED_GEMM"] = "1"
    try:
        return _check_all_pinned()
    finally:
        if before is None:
            del os.environ["ECOFLAP_PINNED_GEMM"]
        else:
            os.environ["ECOFLAP_PINNED_GEMM"] = before


def _check_all_pinned():
    from ecoflap_amd.shapes import fused
    import torch.nn.functional as F
    report = {}
    for N, K, dt, has_bias, rows in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(N + K + rows)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(dt)
        b = (torch.randn(N, device="cuda", generator=g) * 0.1).to(dt) if has_bias else None
        x = (torch.randn(16 * rows, K, device="cuda", generator=g) * 0.7).to(dt)
        with torch.no_grad():
            whole = fused.linear(x, w, b)
            assert whole is not None, (N, K, "no pinned solution")
            plan = fused.pinned_plans()[(N, K, dt)]
            name = plan["name"]
            if dt == torch.float32:           # no library solution to pin: the build's own MFMA GEMM
                assert plan["index"] == -1 and "gemm_f32" in name
            else:
                assert plan["index"] >= 0 and plan["passed"] >= 1
                assert "_SK" not in name or "_SK0" in name, name
            ref = F.linear(x.float(), w.float(), None if b is None else b.float())
            err = (whole.float() - ref).abs().max().item()
            tol = 1e-5 if dt == torch.float32 else 2e-2
            assert err <= tol * ref.abs().max().item() + (1e-5 if dt == torch.float32 else 1e-3), (N, K, err)
            for slot in (0, 7, 15):
                alone = fused.linear(x[slot * rows:(slot + 1) * rows].contiguous(), w, b)
                assert torch.equal(alone, whole[slot * rows:(slot + 1) * rows]), (N, K, rows, slot)
            one = fused.linear(x[5:6].contiguous(), w, b)                # a single row
            assert torch.equal(one, whole[5:6]), (N, K, "one row")
            for _ in range(50):
                assert torch.equal(fused.linear(x, w, b), whole), (N, K, "not repeatable")
            # under graph capture (planned, and this M seen eagerly)
            xs = x[:rows].contiguous()
            eager = fused.linear(xs, w, b)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                cap = fused.linear(xs, w, b)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(cap, eager)
        report[f"{N}x{K} {str(dt).split('.')[-1]} bias={has_bias}"] = plan
    return report


def test_pinned_linears_are_batch_invariant_and_repeatable():
    report = _check_all()
    for k, v in report.items():
        print(k, v["index"], f"{v['us_at_16_slots']:.0f} us (library's first choice: "
              f"{v['library_first_choice_us']:.0f} us)", v["name"][:100])


def test_pinned_linears_do_not_depend_on_the_stream_k_variable():
    """A child process with TENSILE_STREAMK_DATA_PARALLEL=0 (the package's `setdefault` leaves an
    explicit value alone): torch's own GEMM choice is then neither batch invariant nor
    reproducible (blas_guard.py), the pinned solutions still are."""
    env = dict(os.environ, TENSILE_STREAMK_DATA_PARALLEL="0", ECOFLAP_ALLOW_STREAMK="1")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "import torch, ecoflap_amd, os; assert os.environ['TENSILE_STREAMK_DATA_PARALLEL'] == '0';"
            "import test_pinned_gemm as t; t._check_all(); print('ok')") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]


def test_pinned_model_forward_equals_its_own_batched_form_at_batch_size_1(monkeypatch):
    """BLIP-2's ViT-g block at batch size 1 (257 rows per evaluation): 16 concatenated evaluations
    == each alone, bit for bit — the case torch's own GEMM choice fails (different kernels for
    M = 257 and M = 4112) and the reason the loop could not share evaluations at batch size 1."""
    from ecoflap_amd.shapes.eva_vit import Block, half_linear_weights
    from ecoflap_amd.shapes.fused import pin_linears
    monkeypatch.setenv("ECOFLAP_PINNED_GEMM", "1")
    torch.manual_seed(0)
    blk = Block(1408, 16, 6144).cuda().eval()
    half_linear_weights(blk)
    assert pin_linears(blk) == 4
    x = (torch.randn(16, 257, 1408, device="cuda") * 0.5).half()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        whole = blk(x, None)
        for i in (0, 9, 15):
            assert torch.equal(blk(x[i:i + 1], None), whole[i:i + 1]), i


def test_eva_qkv_bias_rides_in_the_gemm_epilogue_as_in_the_reference():
    """EVA's q / v biases (the synthetic models leave them at zero: here they are not).  A
    pinned-forward qkv Linear adds cat(q_bias, 0, v_bias) in its GEMM — the reference's own
    `F.linear(x, weight, qkv_bias)` (eva_vit.py:119-141), bit for bit —; the block's output is the
    un-pinned block's (GEMM, then the separate bias pass: one rounding more) within fp16 rounding,
    a changed bias is picked up, and the block stays batch invariant."""
    import torch.nn.functional as F
    from ecoflap_amd.shapes.eva_vit import Block, half_linear_weights
    from ecoflap_amd.shapes.fused import pin_linears
    torch.manual_seed(1)
    blk = Block(1408, 16, 6144).cuda().eval()
    half_linear_weights(blk)
    with torch.no_grad():
        blk.attn.q_bias.normal_(0, 0.3)
        blk.attn.v_bias.normal_(0, 0.3)
    x = (torch.randn(8, 257, 1408, device="cuda") * 0.5).half()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        plain = blk(x, None)                                   # nn.Linear forwards + ecoflap_qkv_bias_add
        assert blk.attn._epilogue_bias(x) is None
        assert pin_linears(blk) == 4
        bias = blk.attn._epilogue_bias(x)
        assert bias is not None and bias.dtype == torch.float16 and blk.attn._epilogue_bias(x) is bias
        want = F.linear(x, blk.attn.qkv.weight, bias)
        blk.attn.qkv._call_bias = bias
        got = blk.attn.qkv(x)
        blk.attn.qkv._call_bias = None
        assert torch.equal(got, want)
        assert not torch.equal(blk.attn.qkv(x), want)          # (without the hand-over: no bias)
        pinned = blk(x, None)
        err = (pinned.float() - plain.float()).abs().max().item()
        assert 0 < err <= 4e-3 * plain.float().abs().max().item(), err
        assert torch.equal(blk(x[3:4], None), pinned[3:4])
        blk.attn.v_bias.add_(1.0)                              # in place: the cache follows the version
        assert blk.attn._epilogue_bias(x) is not bias
        assert not torch.equal(blk(x, None), pinned)
    with torch.enable_grad():
        assert blk.attn._epilogue_bias(x) is None              # autograd: the torch op chain


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_bias_consumers_equal_the_torch_op_chains(dt):
    """The ops that add a deferred Linear bias (shapes/fused.py): `dtype(a + bias)` first — the
    Linear's own output from a rounded accumulator — then the consumer, against the same chain in
    torch ops: residual add and LayerNorm input bit for bit, GELU within one unit in the last
    place of the 16-bit result (two erf implementations)."""
    from ecoflap_amd.shapes import fused
    torch.manual_seed(3)
    rows, d = 1031, 1408
    a = (torch.randn(rows, d, device="cuda") * 0.7).to(dt)
    x = (torch.randn(rows, d, device="cuda") * 0.7).to(dt)
    b = (torch.randn(d, device="cuda") * 0.2).to(dt)
    with torch.no_grad():
        assert torch.equal(fused.bias_add_residual(x, a, b), x + (a + b))
        got = fused.bias_gelu(a, b).float()
        want = torch.nn.functional.gelu((a + b).float()).to(dt).float()
        ulp = torch.finfo(dt).eps * want.abs().clamp_min(torch.finfo(dt).tiny)
        assert ((got - want).abs() <= ulp).all()
        norm = torch.nn.LayerNorm(d, eps=1e-6).cuda()
        torch.nn.init.normal_(norm.weight, 1.0, 0.1)
        torch.nn.init.normal_(norm.bias, 0.0, 0.1)
        with torch.autocast("cuda", dtype=dt):
            s, y = fused.add_layernorm(x, a, norm, residual_bias=b)
            s2, y2 = fused.add_layernorm(x, a + b, norm)
        assert torch.equal(s, x + (a + b)) and torch.equal(s, s2) and torch.equal(y, y2)


def test_pinned_block_defers_its_biases_and_tracks_the_plain_block(monkeypatch):
    """An EVA block with pinned Linears (biases of proj / fc1 / fc2 added by the consuming ops)
    against the same block on torch's own GEMMs: same function to 16-bit rounding noise; and the
    owning-Linear per-slot form of the loop gives the bits of the module's own forward."""
    import copy
    from ecoflap_amd.shapes.eva_vit import Block, half_linear_weights
    from ecoflap_amd.shapes.fused import pin_linears
    torch.manual_seed(1)
    plain = Block(1408, 16, 6144).cuda().eval()
    for p in plain.parameters():
        if p.dim() == 1:
            torch.nn.init.normal_(p, 0.0, 0.05)          # non-zero biases
    plain.norm1.weight.data.add_(1.0)
    plain.norm2.weight.data.add_(1.0)
    half_linear_weights(plain)
    pinned = copy.deepcopy(plain)
    assert pin_linears(pinned) == 4
    assert all(m.__dict__.get("_defer_bias") for m in (pinned.attn.proj, pinned.mlp.fc1, pinned.mlp.fc2))
    x = (torch.randn(8, 257, 1408, device="cuda") * 0.5).half()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        y0 = plain(x, None).float()
        y1 = pinned(x, None).float()
    assert not any(m.__dict__.get("_bias_pending") for m in pinned.modules())      # every bias was taken
    err = (y0 - y1).abs().max().item()
    assert err <= 2e-2 * y0.abs().max().item(), err
    # the deferral is the BLOCK's request, per call (round 5): anyone else calling these Linears —
    # `blk.mlp(x)`, `blk.attn(x)`, a hook on the Linear's output — gets the bias, and no request
    # or pending flag outlives a call
    for m in (pinned.mlp.fc1, pinned.mlp.fc2, pinned.attn.proj):
        m.bias.data.fill_(3.0)
    for m in (plain.mlp.fc1, plain.mlp.fc2, plain.attn.proj):
        m.bias.data.fill_(3.0)
    seen = []
    hook = pinned.mlp.fc2.register_forward_hook(lambda mod, i, o: seen.append(o.float().mean().item()))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for fn in (lambda b: b.mlp(x), lambda b: b.attn(x), lambda b: b.mlp.fc1(x)):
            a0, a1 = fn(plain).float(), fn(pinned).float()
            assert (a0 - a1).abs().max().item() <= 2e-2 * a0.abs().max().item()       # 3.0 apart if dropped
        hook.remove()
        ref_mean = plain.mlp(x).float().mean().item()
    assert seen and abs(seen[0] - ref_mean) < 0.1
    assert not any(m.__dict__.get("_bias_pending") or m.__dict__.get("_defer_now") for m in pinned.modules())


def test_f32_mfma_gemm_is_exact_to_rounding_batch_invariant_and_repeatable(monkeypatch):
    """csrc/gemm_f32.hip (the Q-Former's fp32 Linears: no library solution to pin on gfx950):
    equals torch's fp32 GEMM to fp32 rounding, a row's result is the same bits alone, in a ragged
    problem and in a 16-slot one, and on every repeated call; shapes it does not tile fall back."""
    from ecoflap_amd.shapes import fused
    import torch.nn.functional as F
    monkeypatch.setenv("ECOFLAP_PINNED_GEMM", "1")
    for N, K, rows in [(768, 768, 8 * 32), (3072, 768, 32), (768, 3072, 8 * 32), (768, 1408, 257), (2048, 768, 33)]:
        g = torch.Generator(device="cuda").manual_seed(N + K + rows)
        w = torch.randn(N, K, device="cuda", generator=g) * 0.05
        b = torch.randn(N, device="cuda", generator=g) * 0.1
        x = torch.randn(16 * rows, K, device="cuda", generator=g)
        with torch.no_grad():
            whole = fused.linear(x, w, b)
            assert whole is not None and whole.dtype == torch.float32
            ref = F.linear(x.double(), w.double(), b.double())
            err = (whole.double() - ref).abs().max().item()
            assert err <= 2e-6 * ref.abs().max().item() * (K / 768) ** 0.5 + 1e-6, (N, K, err)
            for slot in (0, 7, 15):
                alone = fused.linear(x[slot * rows:(slot + 1) * rows].contiguous(), w, b)
                assert torch.equal(alone, whole[slot * rows:(slot + 1) * rows]), (N, K, rows, slot)
            ragged = fused.linear(x[3:3 + 130].contiguous(), w, b)          # 130 rows: a partial tile
            assert torch.equal(ragged, whole[3:133])
            nobias = fused.linear(x[:rows].contiguous(), w, None)
            assert torch.allclose(nobias + b, whole[:rows], rtol=0, atol=1e-6 * float(whole.abs().max()))
            for _ in range(20):
                assert torch.equal(fused.linear(x, w, b), whole)
    with torch.no_grad():                                                   # N % 128 != 0: not tiled
        assert fused.linear(torch.randn(8, 64, device="cuda"), torch.randn(100, 64, device="cuda"), None) is None
