"""The oracle (oracle/) against the golden vectors the reference itself produced
(tests/golden/make_golden.py).  CPU only.  Integer / bit work: exact.  Float
reductions: 1e-5 relative (north_star allows 1e-4)."""
import os

import numpy as np
import pytest
import torch

from helpers import NP2T, from_bits, to_bits


def test_k1_three_pass_bit_exact(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_zo_perturb.npz"))
    for case in g["cases"]:
        dt_name, n, eps, seed = str(case).split("|")
        dt = NP2T[dt_name]
        key = f"{dt_name}_{n}_{seed}"
        w = from_bits(g[key + "_w0"], dt).clone()
        z = from_bits(g[key + "_z"], dt)
        for step, sf in enumerate([1.0, -2.0, 1.0]):
            oracle.zo_perturb(w, sf, float(eps), z)
            assert np.array_equal(to_bits(w), g[key + f"_w{step + 1}"]), (key, step)
        # fused triple == three passes
        w0 = from_bits(g[key + "_w0"], dt).clone()
        plus, minus, rest = oracle.zo_perturb_triple(w0, float(eps), z)
        assert np.array_equal(to_bits(plus), g[key + "_w1"])
        assert np.array_equal(to_bits(minus), g[key + "_w2"])
        assert np.array_equal(to_bits(rest), g[key + "_w3"])


def test_k1_restore_is_not_exact_in_bf16(golden_dir):
    """the reference's +eps,-2eps,+eps cycle drifts; the goldens carry it."""
    g = np.load(os.path.join(golden_dir, "g1_zo_perturb.npz"))
    key = "bf16_5003_123456789"
    assert (g[key + "_w0"] != g[key + "_w3"]).mean() > 0.2


def test_torch_sum_restatement_matches_torch():
    from oracle import torch_sum_f32
    rng = np.random.default_rng(1)
    for trial in range(400):
        n = int(rng.integers(1, 260))
        x = (rng.random(n) * 10.0 ** rng.integers(-3, 9)).astype(np.float32)
        if trial % 3 == 0:
            x = np.ceil(x)
        assert float(torch_sum_f32(x)) == torch.from_numpy(x).sum().item(), n


def test_allocator_exact(golden_dir):
    from oracle import compute_sparsity_per_group
    g = np.load(os.path.join(golden_dir, "g4_allocator.npz"))
    n = int(g["n"])
    assert n > 150
    for i in range(n):
        sp, mx, keep = g[f"{i}_meta"]
        out, _ = compute_sparsity_per_group(int(keep), g[f"{i}_scores"], g[f"{i}_nums"], float(mx))
        assert np.array_equal(np.array(out, dtype=np.float64), g[f"{i}_out"], equal_nan=True), \
            str(g[f"{i}_tag"])


def test_wrapped_gpt_running_mean(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_wrapped_gpt.npz"))
    for case in g["cases"]:
        key, steps = str(case).split("|")
        dt = NP2T[key.split("_")[0]]
        cols = int(key.split("_")[1])
        s = torch.zeros(cols, dtype=torch.float32)
        n = 0
        for i in range(int(steps)):
            x = from_bits(g[f"{key}_x{i}"], dt)
            b = 1 if x.dim() == 2 else x.shape[0]   # wanda_pruner.py:72-74
            x2 = x.reshape(-1, x.shape[-1]).contiguous()
            oracle.colsqnorm_accum(s, x2, n, b)
            n += b
            # bit for bit: the oracle adds the squares in torch's own order (one fp32 fma chain per
            # column over the tokens)
            assert np.array_equal(s.numpy(), g[f"{key}_s{i}"]), (key, i)


def test_wrapped_gpt_raw_rows_replayed_equal_the_running_mean(oracle, golden_dir):
    """The decomposition the data-parallel stage 2 exchanges: per-batch ||x_c||^2 rows
    (`oracle_colsq_raw`) replayed in order (`oracle_colsq_replay`) == WrappedGPT.add_batch call
    by call (wanda_pruner.py:71-84) — bit for bit against the fused restatement and against the
    reference's own scaler_row goldens."""
    g = np.load(os.path.join(golden_dir, "g5_wrapped_gpt.npz"))
    for case in g["cases"]:
        key, steps = str(case).split("|")
        dt = NP2T[key.split("_")[0]]
        cols = int(key.split("_")[1])
        fused = torch.zeros(cols, dtype=torch.float32)
        rows, bs, n = [], [], 0
        for i in range(int(steps)):
            x = from_bits(g[f"{key}_x{i}"], dt)
            b = 1 if x.dim() == 2 else x.shape[0]
            x2 = x.reshape(-1, x.shape[-1]).contiguous()
            oracle.colsqnorm_accum(fused, x2, n, b)
            n += b
            r = torch.empty(cols, dtype=torch.float32)
            oracle.colsq_raw(r, x2)
            rows.append(r)
            bs.append(b)
            rep = torch.zeros(cols, dtype=torch.float32)
            oracle.colsq_replay(rep, torch.stack(rows), bs)
            assert torch.equal(rep, fused), (key, i)
            assert np.array_equal(rep.numpy(), g[f"{key}_s{i}"]), (key, i)


def test_wrapped_gpt_statistic_equals_torch_at_true_sizes(oracle):
    """The reference's op chain (wanda_pruner.py:71-84: `inp.t()`, `.type(torch.float32)`,
    `torch.norm(inp, p=2, dim=1) ** 2 / nsamples`) evaluated by torch on the CPU, against the
    oracle, at the BASELINE shapes' token counts and widths and in their dtypes: bit for bit
    (an ulp here moves Wanda masks at true row lengths: tests/test_true_width.py)."""
    g = torch.Generator().manual_seed(5)
    for dt, tokens, cols, batch in [(torch.bfloat16, 16, 2048, 1), (torch.bfloat16, 8 * 48, 5120, 8),
                                    (torch.float16, 2 * 257, 1408, 2), (torch.float16, 8 * 257, 6144, 8),
                                    (torch.float32, 8 * 197, 768, 8), (torch.float32, 8 * 197, 3072, 8)]:
        row_ref = torch.zeros(cols, dtype=torch.float32)
        row = torch.zeros(cols, dtype=torch.float32)
        n = 0
        for _ in range(2):
            x = (torch.randn(tokens, cols, generator=g) * 1.3).to(dt)
            inp = x.t().type(torch.float32)
            row_ref *= n / (n + batch)
            row_ref += torch.norm(inp, p=2, dim=1) ** 2 / (n + batch)
            oracle.colsqnorm_accum(row, x, n, batch)
            n += batch
            assert torch.equal(row, row_ref), (dt, tokens, cols, int((row != row_ref).sum()))


def test_philox_known_answer(oracle):
    """Random123 known-answer vectors (kat_vectors) for philox4x32 at 7 and 10 rounds:
    all-zero, all-ones and the pi-digits counter/key."""
    kat = [
        (7, [0] * 4, [0] * 2, [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]),
        (10, [0] * 4, [0] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        (7, [0xffffffff] * 4, [0xffffffff] * 2, [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]),
        (10, [0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        (7, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]),
        (10, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for rounds, ctr, key, want in kat:
        assert oracle.philox4x32(ctr, key, rounds) == want, (rounds, ctr)


def philox_rounds():
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "ecoflap_hip.h")).read()
    return int(re.search(r"#define ECOFLAP_PHILOX_ROUNDS (\d+)", text).group(1))


def test_philox_stream_bit_statistics(oracle):
    """The build's word stream (rounds from the header): per-bit frequency, avalanche of a
    one-bit counter change, and independence across adjacent keys."""
    rounds = philox_rounds()
    assert rounds in (7, 10)
    n = 1 << 16
    w = oracle.philox_u32(n, 123456789, rounds).numpy().astype(np.uint32)
    bits = ((w[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.float64)
    assert np.abs(bits.mean(0) - 0.5).max() < 4.5 * 0.5 / np.sqrt(n)           # monobit, each position
    a = np.array([oracle.philox4x32([i, 0, 0, 0], [5, 6], rounds) for i in range(512)], dtype=np.uint32)
    b = np.array([oracle.philox4x32([i ^ 1, 0, 0, 0], [5, 6], rounds) for i in range(512)], dtype=np.uint32)
    flips = np.unpackbits((a ^ b).view(np.uint8)).mean()
    assert abs(flips - 0.5) < 0.01                                               # avalanche
    w2 = oracle.philox_u32(n, 123456790, rounds).numpy().astype(np.uint32)       # next key
    x = w.astype(np.float64) / 2**32 - 0.5
    y = w2.astype(np.float64) / 2**32 - 0.5
    assert abs(np.corrcoef(x, y)[0, 1]) < 0.02
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 0.02


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_normal_stream_restatement_properties(oracle, dt):
    """oracle_normal_stream (the build's in-register z, restated): prefix-stable in n, N(0,1)
    moments, independent across seeds, and the documented element layout — the two outputs of a
    Box-Muller pair satisfy z_cos^2 + z_sin^2 = -2 ln(u1) with u1 from the pair's radius word."""
    rounds = philox_rounds()
    n = 1 << 18
    z, z32 = oracle.normal_stream(n, dt, 42, rounds, want_f32=True)
    assert torch.equal(z[:1003], oracle.normal_stream(1003, dt, 42, rounds))
    assert torch.equal(z[:1024 + 11], oracle.normal_stream(1024 + 11, dt, 42, rounds))
    zd = z32.double()
    assert abs(zd.mean().item()) < 8e-3 and abs(zd.var().item() - 1) < 1e-2
    assert abs((zd ** 4).mean().item() - 3) < 8e-2
    assert abs((zd[:-1] * zd[1:]).mean().item()) < 8e-3
    other = oracle.normal_stream(n, dt, 43, rounds, want_f32=True)[1].double()
    assert abs((zd * other).mean().item()) < 8e-3
    if dt != torch.float32:
        assert torch.equal(z, z32.to(dt))                       # one rounding to the storage dtype
    # layout: first Box-Muller pair and its radius word
    words = oracle.philox_u32(12, 42, rounds).numpy().astype(np.uint32)
    u1 = np.float32(np.float32(words[0]) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33))
    r2 = -2.0 * np.log(float(u1))
    cos_i, sin_i = (0, 1) if dt == torch.float32 else (0, 2)     # N32: (cos, sin); N16: (cosA, cosB, sinA, sinB)
    assert abs(float(z32[cos_i]) ** 2 + float(z32[sin_i]) ** 2 - r2) < 1e-5 * max(1.0, r2)
    if dt != torch.float32:
        ang = (int(words[8]) & 0xffff) / 65536.0                 # 16-bit angle, low half of W[8]
        assert abs(float(z32[0]) - np.sqrt(r2) * np.cos(2 * np.pi * ang)) < 1e-5
        # second vector of block 0 is vector 64: elements 512..519 come from pairs 4..7
        u1b = np.float32(np.float32(words[4]) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33))
        assert abs(float(z32[512]) ** 2 + float(z32[514]) ** 2 + 2.0 * np.log(float(u1b))) < 1e-4


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_oracle_k1_chain_on_every_finite_16_bit_weight_equals_torchs_cpu_ops(oracle, dt):
    """The oracle's restatement of `param.data = param.data + scaling_factor * z * zo_eps`
    (layer_single_base_pruner.py:485-486) against the expression itself, evaluated by torch on the
    CPU as the reference evaluates it: EVERY finite 16-bit weight value x 48 draws of z (incl. +-0,
    the largest normals a run draws and subnormal products), scaling factors +1 / -2 / +1 in place
    and the three-output form.  (The GPU twin of this test holds the HIP kernels to torch's ops on
    the device: tests/test_torch_stream.py.)"""
    allbits = torch.arange(-32768, 32768, dtype=torch.int32).to(torch.int16).view(dt)
    finite = allbits[torch.isfinite(allbits.float())]
    g = torch.Generator().manual_seed(5)
    zs = torch.randn(44, generator=g).tolist() + [0.0, -0.0, 6.5, -6.25]
    z = torch.tensor(zs).to(dt).repeat_interleave(finite.numel())
    w0 = finite.repeat(len(zs)).contiguous()
    eps = 1e-3
    ref = w0.clone()
    got = w0.clone()
    for sf in (1.0, -2.0, 1.0):
        ref = ref + sf * z * eps                              # the reference's expression, torch CPU
        oracle.zo_perturb(got, sf, eps, z)
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), sf
    plus, minus, rest = oracle.zo_perturb_triple(w0.clone(), eps, z)
    rp = w0 + 1.0 * z * eps
    rm = rp + -2.0 * z * eps
    rr = rm + 1.0 * z * eps
    for a, b in ((plus, rp), (minus, rm), (rest, rr)):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_oracle_wanda_selection_fuzz_equals_the_references_expressions():
    """180 random cases of the three Wanda selections, the oracle against the reference's own torch
    expressions evaluated on the CPU (wanda_pruner.py:260-279 rows: stable sort + scatter; :541-558
    matrix: `sort(metric.flatten())[k]`, `metric <= thres`; :265-270 n:m: topk per group): ties,
    zeros, heavy tails, three dtypes, k from 0 to the row / matrix size."""
    from oracle_backend import OracleKernels
    ok = OracleKernels()
    g = torch.Generator().manual_seed(99)
    for case in range(180):
        rows = int(torch.randint(1, 40, (1,), generator=g))
        cols = int(torch.randint(1, 30, (1,), generator=g)) * 4
        dt = [torch.float32, torch.float16, torch.bfloat16][case % 3]
        kind = ["normal", "ties", "zeros", "heavy"][case % 4]
        w = torch.randn(rows, cols, generator=g) * 0.02
        s = torch.rand(cols, generator=g) + 0.05
        if kind == "ties":
            w = torch.round(w * 50) / 50
            s = torch.full((cols,), 0.25)
        elif kind == "zeros":
            w = w * (torch.rand(rows, cols, generator=g) < 0.4)
        elif kind == "heavy":
            w = w * torch.exp(2.5 * torch.randn(rows, cols, generator=g))
        w = w.to(dt)
        metric = torch.abs(w) * torch.sqrt(s.reshape((1, -1)))
        frac = [0.5, 0.37, 0.9, 0.0, 0.6, 1.0, 0.25][case % 7]
        mode = case % 3 if kind == "normal" else case % 2          # (n:m: topk's tie order is not the subject)
        if mode == 0:                                              # rows
            k = int(cols * frac)
            mask = torch.zeros_like(metric) == 1
            mask.scatter_(1, torch.sort(metric, dim=-1, stable=True)[1][:, :k], True)
            got = w.clone()
            ok.wanda_prune_rows(got, s, k)
        elif mode == 1:                                            # matrix
            k = min(rows * cols - 1, int(rows * cols * frac))
            thres = torch.sort(metric.flatten())[0][k]
            mask = metric <= thres
            got = w.clone()
            ok.wanda_prune_matrix(got, s, k)
        else:                                                      # 2:4
            mask = torch.zeros_like(metric) == 1
            for ii in range(0, cols, 4):
                tmp = metric[:, ii:ii + 4].float()
                mask.scatter_(1, ii + torch.topk(tmp, 2, dim=1, largest=False)[1], True)
            got = w.clone()
            ok.wanda_prune_nm(got, s, 2, 4)
        want = w.clone()
        want[mask] = 0
        assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), (case, rows, cols, str(dt), kind, mode)


def test_oracle_k6_fuzz_equals_the_references_add_batch_on_the_cpu():
    """120 random sequences of `WrappedGPT.add_batch` calls (wanda_pruner.py:71-84: 1-3 batches of
    1-4 samples x 1-300 tokens x 2-700 columns, three dtypes): the oracle's running statistic equals
    the reference's expression evaluated by torch on the CPU — `scaler_row *= n / (n + b)`,
    `+= torch.norm(inp.float(), p=2, dim=1) ** 2 / n` — float bit for float bit, i.e. the oracle sums
    in torch's own CPU order.  (One column is the one shape where it does not: torch reduces a single
    contiguous row with another kernel; no Linear has one input feature.)"""
    from oracle_backend import OracleKernels
    ok = OracleKernels()
    g = torch.Generator().manual_seed(3)
    for case in range(120):
        cols = int(torch.randint(2, 700, (1,), generator=g))
        dt = [torch.float32, torch.float16, torch.bfloat16][case % 3]
        srow, ref, n = torch.zeros(cols), torch.zeros(cols), 0
        for _ in range(int(torch.randint(1, 4, (1,), generator=g))):
            B = int(torch.randint(1, 5, (1,), generator=g))
            T = int(torch.randint(1, 300, (1,), generator=g))
            x = (torch.randn(B, T, cols, generator=g) * 1.7).to(dt)
            inp = x.reshape((-1, cols)).t()
            ref *= n / (n + B)
            ref += torch.norm(inp.type(torch.float32), p=2, dim=1) ** 2 / (n + B)
            ok.colsqnorm_accum(srow, x.reshape(-1, cols), n, B)
            n += B
        assert torch.equal(ref.view(torch.int32), srow.view(torch.int32)), (case, cols, str(dt))
