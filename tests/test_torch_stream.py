"""The reference's own draw inside K1 (round 5): `torch.manual_seed(seed); torch.normal(0, 1, size,
device, dtype)` (layer_single_base_pruner.py:482-485) regenerated in registers by the HIP kernels.

GPU tests: the regenerated stream equals torch.normal of THIS torch on THIS device bit for bit
(three dtypes, the five BLIP-2 weight shapes, ragged sizes); every K1 form fed that stream equals
the oracle's three-rounding chain fed torch's own tensor; a whole stage-1 run in registers equals
the run that materialises every z.  CPU tests: the thread-count rule of ATen's launch."""
import os

import numpy as np
import pytest
import torch

DTYPES = [torch.float32, torch.float16, torch.bfloat16]


def bits(t):
    return t.contiguous().view(torch.int32 if t.dtype == torch.float32 else torch.int16)


def test_thread_rule_of_atens_launch():
    """ecoflap_torch_normal_threads == calc_execution_policy (ATen/native/cuda/
    DistributionTemplates.h:50-62): 256 * min(SMs * (maxThreadsPerSM / 256), ceil(n / 256))."""
    from ecoflap_amd import hip
    lib = hip.load_library()
    f = lib.ecoflap_torch_normal_threads
    for n in (1, 255, 256, 257, 1003, 589824, 4194304, 10485760, 1 << 33):
        for sms, mt in ((256, 2048), (304, 2048), (108, 1536)):
            want = 256 * min(sms * (mt // 256), -(-n // 256))
            assert f(n, sms, mt) == want, (n, sms, mt)
    assert f(0, 256, 2048) == 0 and f(5, 0, 2048) == 0 and f(5, 256, 128) == 0
    # argument checks of the launch functions (no GPU needed: they fail before any launch)
    vp = __import__("ctypes").c_void_p
    assert lib.ecoflap_zo_fill_normal_torch(vp(16), 1024, 1, 1, 100, None) == -3      # threads % 256
    assert lib.ecoflap_zo_fill_normal_torch(vp(16), 1024, 1, 1, 2048, None) == -3     # > ceil(n/256)*256
    assert lib.ecoflap_zo_fill_normal_torch(None, 1024, 1, 1, 1024, None) == -2
    # 2^31 elements or more: torch draws such a tensor in several launches (with_32bit_indexing),
    # each at its own Philox offset — refused here, LayerSparsity materialises that draw
    assert lib.ecoflap_zo_fill_normal_torch(vp(16), 2 ** 31, 1, 1, 256 * 2048, None) == -3
    assert lib.ecoflap_zo_perturb_torch(vp(16), 2 ** 31 + 8, 1, 1.0, 1e-3, 1, 256 * 2048, None) == -3
    assert lib.ecoflap_zo_fill_normal_torch(vp(8), 1024, 1, 1, 1024, None) == -5
    assert lib.ecoflap_zo_perturb_torch(vp(16), 1024, 7, 1.0, 1e-3, 1, 1024, None) == -1
    assert lib.ecoflap_zo_perturb_layers_torch(None, 1, 1, 1, 1e-3, None, None, None) == -2
    assert lib.ecoflap_zo_perturb_layers_torch(vp(16), 1, 1, 1, 1e-3, None, vp(1), None) == -2


def test_layer_items_count_the_items_that_hold_a_vector():
    """ecoflap_torch_layer_items: full rounds take ceil(T / N / 64) items, the last round only the
    wave chunks that hold a vector of its first row (the items behind them are empty), at least
    one — against a brute-force count over the lane geometry of csrc/zo_perturb.hip (item I =
    round j * wpr + chunk c owns vectors (4j + ii) * vpr + 64c + lane, ii = 0..3)."""
    from ecoflap_amd import hip
    lib = hip.load_library()
    f = lib.ecoflap_torch_layer_items
    T_full = 256 * 2048
    for dt, N in ((1, 8), (2, 8), (0, 4)):
        for n in (1, 3, 8, 255, 1003, 4099, T_full - 8, T_full + 3, 4 * T_full + 4104, 1408 * 1408, 4224 * 1408,
                  6144 * 1408, 2048 * 2048, 5120 * 2048, 768 * 768, 2304 * 768 + 5):
            T = 256 * min(2048, -(-n // 256))
            vpr, nvec = T // N, n // N
            wpr = -(-vpr // 64)
            rounds = -(-n // (4 * T))
            last = 0                                   # highest item with any valid vector, + 1
            for j in range(rounds):
                for c in range(wpr):
                    if 64 * c < vpr and 4 * j * vpr + 64 * c < nvec:      # row 0, lane 0
                        last = j * wpr + c + 1
            assert f(n, T, dt) == max(last, 1), (n, dt)
            assert f(n, T, dt) <= rounds * wpr
    assert f(0, 256, 1) == 0 and f(1024, 100, 1) == 0 and f(1024, 1024, 9) == 0


def test_radius_sweep_argument_checks():
    """ecoflap_zo_torch_radius_sweep refuses ranges outside the 2^32 words before any launch."""
    from ecoflap_amd import hip
    lib = hip.load_library()
    vp = __import__("ctypes").c_void_p
    assert lib.ecoflap_zo_torch_radius_sweep(1 << 32, 1, vp(16), None) == -3
    assert lib.ecoflap_zo_torch_radius_sweep(5, 1 << 32, vp(16), None) == -3
    assert lib.ecoflap_zo_torch_radius_sweep(0, 16, None, None) == -2
    assert lib.ecoflap_zo_torch_radius_sweep(7, 0, None, None) == 0


@pytest.fixture(scope="module")
def kern():
    from ecoflap_amd import hip
    assert torch.cuda.is_available()
    return hip.HipKernels()


BLIP2_SHAPES = [(4224, 1408), (1408, 1408), (6144, 1408), (2048, 2048), (5120, 2048)]
RAGGED = [1, 3, 8, 255, 256, 257, 1003, 4099, 256 * 2048 - 8, 256 * 2048 + 3, 4 * 256 * 2048 + 4104,
          768 * 768, 2304 * 768 + 5]


@pytest.mark.gpu
def test_short_radius_sequence_equals_rocrands_on_every_32_bit_word(kern):
    """The Box-Muller radius the kernels compute — two-instruction ln 2 product, v_rsq_f32 and one
    Newton step on an exact residual, a v_max against +0 for the 128 words that round to u = 1
    (csrc/zo_perturb.hip: torch_radius) — equals rocRAND's sqrtf(-2 logf(u)) as compiled into
    torch's kernel (ocml's compensated logf product, correctly rounded sqrtf: torch_radius_reference)
    for EVERY 32-bit word: the function's whole domain, enumerated on the device.  Bit for bit; the
    shortcut is exact here, not in general, and this is its proof."""
    assert kern.torch_radius_sweep(0, 1 << 32) == 0
    # the words that round to u = 1 (x2 = 0), the first word, a mid-range slice
    assert kern.torch_radius_sweep((1 << 32) - 256, 256) == 0
    assert kern.torch_radius_sweep(0, 1) == 0
    assert kern.torch_radius_sweep(0x7fff0000, 1 << 17) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dt", DTYPES)
def test_regenerated_stream_equals_torch_normal(kern, dt):
    assert kern.torch_stream_matches(torch.device("cuda:0")), \
        "the start-up probe itself says this torch's stream is not the one the kernels regenerate"
    for k, shape in enumerate(BLIP2_SHAPES + [(n,) for n in RAGGED]):
        seed = [0, 1, 12341, 999999999, 2 ** 31 + 7][k % 5] + k
        torch.manual_seed(seed)
        want = torch.normal(mean=0, std=1, size=shape, device="cuda", dtype=dt)
        got = torch.empty_like(want)
        kern.zo_fill_normal_torch(got, seed)
        same = torch.equal(bits(want), bits(got))
        if not same:
            d = (bits(want) != bits(got)).flatten().nonzero().flatten()
            raise AssertionError(f"{shape} seed {seed}: {d.numel()} of {want.numel()} differ, first at "
                                 f"{d[:8].tolist()}: {want.flatten()[d[:4]].tolist()} vs "
                                 f"{got.flatten()[d[:4]].tolist()}")


@pytest.mark.gpu
def test_probe_leaves_torch_generators_alone(kern):
    from ecoflap_amd import hip
    fresh = hip.HipKernels()
    torch.manual_seed(77)
    torch.rand(3, device="cuda")
    a_cpu, a_gpu = torch.get_rng_state(), torch.cuda.get_rng_state("cuda:0")
    assert fresh.torch_stream_matches(torch.device("cuda:0"))
    assert torch.equal(a_cpu, torch.get_rng_state())
    assert torch.equal(a_gpu, torch.cuda.get_rng_state("cuda:0"))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n", [1, 5, 8, 1003, 4099, 256 * 2048 + 3, 2048 * 2048, 4 * 256 * 2048 + 4104])
def test_k1_forms_on_torchs_stream_equal_oracle_chain(kern, oracle, dt, n):
    """single / triple / units / layers with z regenerated in registers == the oracle's
    three-rounding arithmetic fed the tensor torch.normal returns for the same seed."""
    from ecoflap_amd.hip import TORCH_Z
    g = torch.Generator().manual_seed(n)
    w0 = (torch.randn(n, generator=g) * 0.05).to(dt)
    seeds = [424242 + n, 17, 2 ** 33 + 5]
    zs = []
    for sd in seeds:
        torch.manual_seed(sd)
        zs.append(torch.normal(mean=0, std=1, size=(n,), device="cuda", dtype=dt).cpu())
    # single: the reference's three in-place passes
    w, ref = w0.clone().cuda(), w0.clone()
    for sf in (1.0, -2.0, 1.0):
        kern.zo_perturb(w, sf, 1e-3, seeds[0], TORCH_Z)
        oracle.zo_perturb(ref, sf, 1e-3, zs[0])
        assert torch.equal(bits(w.cpu()), bits(ref)), sf
    # triple (theta+ aliasing the input, as the loop calls it) and the drift-only form
    cur = w0.clone().cuda()
    minus, rest = torch.empty_like(cur), torch.empty_like(cur)
    kern.zo_perturb_triple(cur, cur, minus, rest, 1e-3, seeds[0], TORCH_Z)
    p, m, r = oracle.zo_perturb_triple(w0.clone(), 1e-3, zs[0])
    for a, b in ((cur, p), (minus, m), (rest, r)):
        assert torch.equal(bits(a.cpu()), bits(b))
    src, only = w0.clone().cuda(), torch.empty(n, dtype=dt, device="cuda")
    kern.zo_perturb_triple(src, None, None, only, 1e-3, seeds[0], TORCH_Z)
    assert torch.equal(bits(only.cpu()), bits(r)) and torch.equal(bits(src.cpu()), bits(w0))
    # units (in place, second unit drift-only) and the same through the block-batched call
    chain, ref = [], w0.clone()
    for z in zs:
        p_, m_, ref = oracle.zo_perturb_triple(ref, 1e-3, z)
        chain.append((p_, m_))
    for form in ("units", "layers"):
        w = w0.clone().cuda()
        plus = [torch.empty_like(w), None, torch.empty_like(w)]
        minus = [torch.empty_like(w), None, torch.empty_like(w)]
        if form == "units":
            kern.zo_perturb_units(w, 1e-3, seeds, plus, minus, TORCH_Z)
            final = w
        else:
            final = torch.empty_like(w)
            other = (torch.randn(777, device="cuda") * 0.05).to(dt)
            other_fin = torch.empty_like(other)
            kern.zo_perturb_layers([(other, other_fin, [5], [None], [None], TORCH_Z),
                                    (w, final, seeds, plus, minus, TORCH_Z)], 1e-3)
            assert torch.equal(bits(w.cpu()), bits(w0))             # originals untouched
        assert torch.equal(bits(final.cpu()), bits(ref)), form
        for u in (0, 2):
            assert torch.equal(bits(plus[u].cpu()), bits(chain[u][0])), (form, u)
            assert torch.equal(bits(minus[u].cpu()), bits(chain[u][1])), (form, u)


def _stage1(kernels, z_env, k1_form, dev="cuda"):
    from ecoflap_amd.pruners.layer_sparsity import LayerSparsity
    from ecoflap_amd.pruners.losses import loss_vision_language
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import blip2_toy
    old = os.environ.get("ECOFLAP_TORCH_Z")
    if z_env is None:
        os.environ.pop("ECOFLAP_TORCH_Z", None)
    else:
        os.environ["ECOFLAP_TORCH_Z"] = z_env
    try:
        torch.manual_seed(0)
        model = blip2_toy().eval().to(dev)
        batches = S.image_text_batches(8, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6, device=dev)
        mapping = {k: k.rsplit(".", 2)[0] for k, v in model.named_parameters()
                   if v.dim() == 2 and (".block" in k) and "relative_attention_bias" not in k}
        np.random.seed(42)
        ls = LayerSparsity(model, batches, loss_vision_language, 8, 0.5, 0.6, "MEZO-GradOnly_sum", 1, 1e-3,
                           mapping, kernels=kernels, z_source="torch", k1_form=k1_form)
        table = ls.return_sparsity()
        torch.cuda.synchronize()
        state = (torch.get_rng_state().clone(), torch.cuda.get_rng_state("cuda:0").clone())
        return table, ls.loss_table.copy(), {k: v.detach().cpu() for k, v in model.state_dict().items()}, \
            ls.stats["z_mode"], state
    finally:
        if old is None:
            os.environ.pop("ECOFLAP_TORCH_Z", None)
        else:
            os.environ["ECOFLAP_TORCH_Z"] = old


@pytest.mark.gpu
@pytest.mark.parametrize("k1_form", ["block", "units", "triple", "single"])
def test_stage1_in_registers_equals_materialised_draws(kern, k1_form):
    """Whole zeroth-order pass, default z source: z regenerated in registers == every z drawn by
    torch.normal and read from memory — loss table, sparsity table, drifted weights bit for bit,
    and torch's generators end in the same state."""
    t_r, l_r, w_r, mode_r, st_r = _stage1(kern, None, k1_form)
    t_m, l_m, w_m, mode_m, st_m = _stage1(kern, "materialised", k1_form)
    assert mode_r == "torch-registers" and mode_m == "torch-materialised"
    assert t_r == t_m
    assert np.array_equal(l_r.view(np.uint32), l_m.view(np.uint32))
    for k in w_r:
        assert torch.equal(w_r[k], w_m[k]), k
    assert torch.equal(st_r[0], st_m[0]) and torch.equal(st_r[1], st_m[1])


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_k1_in_registers_on_every_finite_16_bit_weight_equals_the_references_op_chain(kern, dt):
    """K1's in-register arithmetic takes shortcuts (one fused step for theta-, packed adds) that are
    argued, not generic.  Here EVERY finite 16-bit weight value meets 4096 draws of the reference's
    z (2.7e8 pairs per dtype: all 63 488 / 64 512 weight patterns x the whole range of rounded
    normals), the triple form with z regenerated in registers, against the reference's own op
    chain executed by torch on the same GPU (`param.data = param.data + scaling_factor * z * zo_eps`,
    layer_single_base_pruner.py:485-486, three times: +1, -2, +1 — one rounding per torch op)."""
    from ecoflap_amd.hip import TORCH_Z
    allbits = torch.arange(-32768, 32768, dtype=torch.int32, device="cuda").to(torch.int16).view(dt)
    finite = allbits[torch.isfinite(allbits.float())]
    reps = 4096
    w0 = finite.repeat(reps).contiguous()
    eps, seed = 1e-3, 20260101
    torch.manual_seed(seed)
    z = torch.normal(mean=0, std=1, size=w0.shape, device="cuda", dtype=dt)
    ref_plus = w0 + 1.0 * z * eps
    ref_minus = ref_plus + -2.0 * z * eps
    ref_rest = ref_minus + 1.0 * z * eps
    plus, minus, rest = (torch.empty_like(w0) for _ in range(3))
    kern.zo_perturb_triple(w0, plus, minus, rest, eps, seed, TORCH_Z)
    for name, a, b in (("theta+", plus, ref_plus), ("theta-", minus, ref_minus), ("restored", rest, ref_rest)):
        diff = bits(a) != bits(b)
        n = int(diff.sum())
        if n:
            i = diff.nonzero().flatten()[:4]
            raise AssertionError(f"{name}: {n} of {a.numel()} differ, e.g. w={w0[i].tolist()} z={z[i].tolist()} "
                                 f"got {a[i].tolist()} want {b[i].tolist()}")


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_block_batched_k1_on_every_finite_16_bit_weight_equals_the_references_op_chain(kern, dt):
    """The production form (`ecoflap_zo_perturb_layers_torch`: one launch for a block's layers, U
    units chained on the drifting weights, z regenerated in registers) on every finite 16-bit weight
    value x 1024 positions, two layers of different sizes in one launch, three units: theta+ /
    theta- of every unit and the drifted weights equal the reference's op chain evaluated by torch
    on the same GPU with torch's own draws."""
    from ecoflap_amd.hip import TORCH_Z
    allbits = torch.arange(-32768, 32768, dtype=torch.int32, device="cuda").to(torch.int16).view(dt)
    finite = allbits[torch.isfinite(allbits.float())]
    eps = 1e-3
    layers, want = [], []
    for li, reps in enumerate((1024, 37)):
        w0 = finite.repeat(reps)[: finite.numel() * reps - 5 * li].contiguous()     # (a ragged tail on the second)
        seeds = [901 + 10 * li, 17 + li, 2 ** 33 + 5 + li]
        plus = [torch.empty_like(w0) for _ in seeds]
        minus = [torch.empty_like(w0) for _ in seeds]
        final = torch.empty_like(w0)
        layers.append((w0, final, seeds, plus, minus, TORCH_Z))
        cur, chain = w0.clone(), []
        for sd in seeds:
            torch.manual_seed(sd)
            z = torch.normal(mean=0, std=1, size=w0.shape, device="cuda", dtype=dt)
            p_ = cur + 1.0 * z * eps
            m_ = p_ + -2.0 * z * eps
            cur = m_ + 1.0 * z * eps
            chain.append((p_, m_))
        want.append((chain, cur))
    kern.zo_perturb_layers(layers, eps)
    for (w0, final, seeds, plus, minus, _), (chain, cur) in zip(layers, want):
        assert torch.equal(bits(final), bits(cur))
        for u in range(len(seeds)):
            assert torch.equal(bits(plus[u]), bits(chain[u][0])), u
            assert torch.equal(bits(minus[u]), bits(chain[u][1])), u


@pytest.mark.gpu
def test_k1_in_registers_on_random_fp32_bit_patterns_equals_the_references_op_chain(kern):
    """fp32 cannot be enumerated: 2^26 weights drawn as random 32-bit PATTERNS (every exponent,
    subnormals, huge values; non-finite ones dropped) through the triple form with z regenerated in
    registers, against the reference's op chain evaluated by torch on the same GPU."""
    from ecoflap_amd.hip import TORCH_Z
    g = torch.Generator(device="cuda").manual_seed(8)
    raw = torch.randint(-2 ** 31, 2 ** 31 - 1, (1 << 26,), device="cuda", dtype=torch.int64, generator=g).to(torch.int32)
    w0 = raw.view(torch.float32)
    w0 = w0[torch.isfinite(w0)].contiguous()
    eps, seed = 1e-3, 424243
    torch.manual_seed(seed)
    z = torch.normal(mean=0, std=1, size=w0.shape, device="cuda", dtype=torch.float32)
    rp = w0 + 1.0 * z * eps
    rm = rp + -2.0 * z * eps
    rr = rm + 1.0 * z * eps
    plus, minus, rest = (torch.empty_like(w0) for _ in range(3))
    kern.zo_perturb_triple(w0, plus, minus, rest, eps, seed, TORCH_Z)
    for name, a, b in (("theta+", plus, rp), ("theta-", minus, rm), ("restored", rest, rr)):
        n = int((bits(a) != bits(b)).sum())
        assert n == 0, f"{name}: {n} of {a.numel()} differ"
