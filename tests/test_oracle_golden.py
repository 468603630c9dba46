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
            np.testing.assert_allclose(s.numpy(), g[f"{key}_s{i}"], rtol=1e-5, atol=0)


def test_philox_known_answer(oracle):
    """Random123 known-answer vectors for philox4x32-10 (kat_vectors: counter/key all zero,
    and all ones)."""
    import ctypes
    out = (ctypes.c_uint32 * 4)()
    oracle.lib.oracle_philox4x32_10.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
    oracle.lib.oracle_philox4x32_10(out, 0, 0)
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
